"""Differential soak at sizes the oracle does not reach: the fast paths of one context (barcode table sized from a sample, LDS
table of hot barcodes, key layout planned from a sample, splitter sort with partitions by reservation and the one-atomic ranking,
matrices over PCIe as bytes widened on host threads) against the conservative paths
of the same library (exact ingest statistics in cb_insert, no hot list, LSD sort) on random large streams: every observable equal."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
import numpy as np
from dropest_amd import capi
from dropest_amd.synth import SynthStream, inject_n

OLD = {"DROPEST_EXACT_INGEST_STATS": "1", "DROPEST_CB_NO_HOT": "1", "DROPEST_SORT": "lsd", "DROPEST_SS_BALLOT_RANK": "1",
       # round 5: no fused key pass / (cell, gene) rows out of the compaction (both moot under the LSD sort), the merge folds with seg_reduce twice,
       # the filtered cells are ordered on the host
       "DROPEST_NO_FUSED_KEYS": "1", "DROPEST_SS_NO_FUSED_CG": "1", "DROPEST_NO_FUSED_FOLD": "1", "DROPEST_SORTF_HOST": "1"}
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dropest_amd", "data", "barcodes")


def run(dev, side, kw, env):
    for k in OLD:
        os.environ.pop(k, None)
    os.environ.update(env)
    c = capi.Context(**kw)
    if side:
        c.set_side_strings(side)
    c.set_matrix_wire(not env)          # round 4: the conservative side copies the 32-bit matrices as they are; the fast side sends bytes and widens on host threads
    c.push_reads_device(*dev.ptrs, dev.n, adopt=True)
    c.set_initialized(); c.merge_and_filter()
    rows = c.cell_rows()
    out = {"cm": [x.copy() for x in c.count_matrix_csc(filtered=True)], "raw": [x.copy() for x in c.count_matrix_csc(filtered=False)],
           "rows": {k: rows[k].copy() for k in rows.dtype.names}, "filtered": np.array(c.filtered_cells()), "targets": np.array(c.merge_targets()),
           "counters": list(c.global_counters()), "layout": c.sort_layout()}
    c.close()
    return out


rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
for it in range(int(os.environ.get("ITERS", "6"))):
    n = int(rng.integers(5_000_000, int(os.environ.get("NMAX", "40000000"))))
    merge = bool(rng.integers(0, 2))
    shape = dict(n_reads=n, n_cells=int(rng.integers(50, 20_000)), n_genes=int(rng.integers(200, 40_000)), umi_len=int(rng.choice([8, 10, 12])),
                 stream_id=int(rng.integers(1, 1000)), permille_neighbour=int(rng.integers(0, 200)) if merge else 50)
    s = SynthStream(**shape)
    dev = s.generate_device(0)
    kw = dict(min_genes_before_merge=int(rng.integers(1, 30)), min_genes_after_merge=int(rng.integers(30, 120)))
    if merge:
        kw.update(merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=capi.BARCODES_CONST, barcodes_file=os.path.join(DATA, "10x_aug_2016_split"))
    t0 = time.time()
    if it < int(os.environ.get("SKIP", "0")):
        dev.free(); continue
    print(it, shape, kw.get("merge_kind"), flush=True)
    a = run(dev, (), kw, OLD)
    only = os.environ.get("NEW_ENV")           # e.g. "DROPEST_CB_NO_HOT=1,DROPEST_SORT=lsd": bisect which fast path differs
    b = run(dev, (), kw, dict(x.split("=") for x in only.split(",")) if only else {})
    for name in ("cm", "raw"):
        for j, (x, y) in enumerate(zip(a[name], b[name])):
            if not np.array_equal(x, y):
                print("DIFF", name, j, len(x), len(y), "first at", int(np.flatnonzero(x[:min(len(x), len(y))] != y[:min(len(x), len(y))])[0]) if len(x) and len(y) else -1,
                      "rows differ:", [k for k in a["rows"] if not np.array_equal(a["rows"][k], b["rows"][k])], a["layout"], b["layout"], flush=True)
                raise SystemExit(1)
    for k in a["rows"]:
        assert np.array_equal(a["rows"][k], b["rows"][k]), (it, k)
    assert np.array_equal(a["filtered"], b["filtered"]) and np.array_equal(a["targets"], b["targets"]) and a["counters"] == b["counters"]
    print(it, shape, "merge" if merge else "", "old", a["layout"]["sort"], "new", b["layout"]["sort"], "cells", len(a["rows"]["barcode"]),
          "nnz", len(a["cm"][1]), "ok %.1fs" % (time.time() - t0), flush=True)
    dev.free()
