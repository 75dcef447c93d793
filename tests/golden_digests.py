"""Golden fixtures of the Estimation path: SHA-256 digests of the canonical observables (cell table, filtered cells,
merge targets, molecule table, both count matrices, per-chromosome rows) of seeded synthetic streams, computed ONCE
by the CPU oracle and committed as tests/golden/digests.json.

    python tests/golden_digests.py --write        # regenerate the fixture (oracle only; no GPU)

tests/test_golden.py checks the oracle against the file on CPU (guards the oracle against drift) and the HIP path
against the same file on the GPU."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
DATA = os.path.join(ROOT, "dropest_amd", "data", "barcodes")
PATH = os.path.join(HERE, "golden", "digests.json")

# name -> (SynthStream arguments, N rate in UMIs, container configuration)
CASES = {
    "c2_1e5": (dict(n_reads=100_000, n_cells=40, n_genes=3000, umi_len=10, stream_id=2), 0.0,
               dict(merge="none", min_before=20, min_after=100)),
    "c2_1e6_n": (dict(n_reads=1_000_000, n_cells=100, n_genes=5000, umi_len=10, stream_id=2), 1e-3,
                 dict(merge="none", min_before=20, min_after=100)),
    "c3_2e5_merge": (dict(n_reads=200_000, n_cells=30, n_genes=2000, umi_len=12, permille_neighbour=150, stream_id=3), 0.0,
                     dict(merge="real", whitelist="10x_aug_2016_split", min_before=3, min_after=20)),
    "c4_2e5_indrop": (dict(n_reads=200_000, n_cells=30, n_genes=2000, umi_len=8, permille_neighbour=150, whitelist="indrop_v3",
                           stream_id=4), 0.0,
                      dict(merge="real", whitelist="indrop_v3", min_before=3, min_after=20)),
    "simple_1e5": (dict(n_reads=100_000, n_cells=25, n_genes=1000, umi_len=8, permille_neighbour=150, stream_id=5), 0.0,
                   dict(merge="simple", max_ed=2, min_before=3, min_after=10)),
    "directional_1e5": (dict(n_reads=100_000, n_cells=30, n_genes=300, umi_len=5, stream_id=6), 0.0,
                        dict(merge="none", umi="directional", min_before=5, min_after=5)),
    # -M (PoissonTargetEstimator): with the 10x whitelist and without; merge_type = all; UMI quality sums through a merge
    "poisson_real_2e5": (dict(n_reads=200_000, n_cells=30, n_genes=2000, umi_len=12, permille_neighbour=150, stream_id=7), 0.0,
                         dict(merge="poisson_real", whitelist="10x_aug_2016_split", min_before=3, min_after=20)),
    "poisson_simple_1e5": (dict(n_reads=100_000, n_cells=25, n_genes=1000, umi_len=8, permille_neighbour=150, stream_id=8), 0.0,
                           dict(merge="poisson_simple", max_ed=2, min_before=3, min_after=10)),
    "merge_all_1e5": (dict(n_reads=100_000, n_cells=25, n_genes=1000, umi_len=8, permille_neighbour=150, stream_id=9), 0.0,
                      dict(merge="all", max_ed=2, min_before=3, min_after=10)),
    "quality_1e5_merge": (dict(n_reads=100_000, n_cells=25, n_genes=60, umi_len=6, permille_neighbour=200, stream_id=10), 1e-3,
                          dict(merge="real", whitelist="10x_aug_2016_split", min_before=3, min_after=10, quality=6)),
}


def qualities(n, qlen):
    """The (seeded) UMI qualities of the cases that carry them: uint8 [n, qlen], phred+33 characters."""
    return np.random.default_rng(4242).integers(33, 75, size=(n, qlen), dtype=np.uint8)


def stream(case):
    import parity
    from dropest_amd.synth import SynthStream, inject_n
    kw, n_rate, cfg = CASES[case]
    s = SynthStream(**kw)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    side = ()
    if n_rate:
        umi, side = inject_n(umi, gene, n_rate, 7, kw["umi_len"])
    return cb, umi, gene, aux, side, cfg


def _h(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def digests(view):
    """view: dict of canonical arrays / lists (see oracle_view / tests/test_golden.py::hip_view)."""
    out = {k: _h(np.asarray(v)) for k, v in view.items() if not k.startswith("_")}
    out["n_cells"] = int(view["_n_cells"]); out["n_real"] = int(view["_n_real"]); out["nnz_cm"] = int(len(view["cm"][0]))
    return out


def canonical(barcodes, rows8, filtered, merge_targets, counters, cm, cm_raw, chr_rows, molecules, quality=None):
    """Everything as fixed-width numpy data.  rows8 = [merged, excluded, real, n_genes, req_genes, req_umis, total_reads,
    total_umis] per cell (sizes of merged source cells zeroed: the reference keeps stale values nobody reads)."""
    rows8 = np.array(rows8, np.int64)
    rows8[rows8[:, 0] != 0, 3:6] = 0
    mol = sorted(molecules)
    extra = {}
    if quality is not None:   # {(cell, gene, umi): sums} -> rows in the molecule table's order
        extra["quality"] = np.array([quality[(m[0], m[1], m[2])] for m in mol], np.int64)
    return {**extra,
        "cells": np.array([b.encode() for b in barcodes], "S40"), "rows": rows8,
        "filtered": np.array(filtered, np.int64), "merge_targets": np.array(merge_targets, np.int64),
        "counters": np.array(counters, np.int64),
        "cm": np.stack([np.asarray(x, np.int64) for x in cm]), "cm_raw": np.stack([np.asarray(x, np.int64) for x in cm_raw]),
        "chr": np.stack([np.asarray(x, np.int64) for x in chr_rows]),
        "molecules": np.array([("%d|%d|%s|%d|%d" % m).encode() for m in mol], "S64"),
        "_n_cells": len(barcodes), "_n_real": int(rows8[:, 2].sum()),
    }


def oracle_view(case):
    import ctypes
    import parity
    from oracle import Oracle
    cb, umi, gene, aux, side, cfg = stream(case)
    kw = dict(min_genes_before=cfg["min_before"], min_genes_after=cfg["min_after"])
    if cfg["merge"] == "real":
        kw.update(merge_kind=1, barcodes_kind=1, barcodes_file=os.path.join(DATA, cfg["whitelist"]))
    elif cfg["merge"] == "simple":
        kw.update(merge_kind=2, max_cb_merge_ed=cfg["max_ed"])
    elif cfg["merge"] == "poisson_real":
        kw.update(merge_kind=3, barcodes_kind=1, barcodes_file=os.path.join(DATA, cfg["whitelist"]))
    elif cfg["merge"] == "poisson_simple":
        kw.update(merge_kind=4, max_cb_merge_ed=cfg["max_ed"])
    elif cfg["merge"] == "all":
        kw.update(merge_kind=5, max_cb_merge_ed=cfg["max_ed"])
    if cfg.get("umi") == "directional":
        kw.update(umi_merge_kind=1)
        ctypes.CDLL("libc.so.6").srand(1)
    qlen = cfg.get("quality")
    if qlen:
        o = Oracle(**kw)
        o.add_packed_q(cb, umi, gene, aux, qualities(len(cb), qlen), side)
        o.set_initialized(); o.merge_and_filter()
    else:
        o = parity.oracle_run(Oracle, kw, cb, umi, gene, aux, side)
    rows = o.cell_rows()
    oc, og, ou, orr, om = o.molecules()
    keep = rows[oc.astype(np.int64), 0] == 0
    mols = [(int(c), int(g), u, int(r), int(m)) for c, g, u, r, m, k in zip(oc, og, ou, orr, om, keep) if k]
    quality = None
    if qlen:
        oq = o.molecule_qualities(len(oc), qlen)
        quality = {(int(c), int(g), u): [int(x) for x in q] for c, g, u, q, k in zip(oc, og, ou, oq, keep) if k}
    return canonical([o.cell_barcode(i) for i in range(o.n_cells)], rows, o.filtered_cells(), o.merge_targets(), o.global_counters(),
                     o.count_matrix(filtered=True), o.count_matrix(filtered=False), o.chr_stats(), mols, quality)


if __name__ == "__main__":
    if "--write" not in sys.argv:
        raise SystemExit(__doc__)
    out = {case: digests(oracle_view(case)) for case in CASES}
    os.makedirs(os.path.dirname(PATH), exist_ok=True)
    with open(PATH, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", PATH, {k: (v["n_cells"], v["n_real"], v["nnz_cm"]) for k, v in out.items()})
