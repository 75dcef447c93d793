"""Soak (not part of the suites): random mid-size streams with Hamming-1 neighbour barcodes through the whitelist CB
merge (-m and -M), GPU against the oracle.  Run on a GPU box: PYTHONPATH=. python scripts/soak_merge_mid.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from dropest_amd import capi
from dropest_amd.synth import SynthStream
from oracle import Oracle
import parity
DATA = os.path.join(os.path.dirname(os.path.abspath(capi.__file__)), "data", "barcodes")
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "77")))
for it in range(int(os.environ.get("SOAK_CASES", "10"))):
    wl = ["10x_aug_2016_split", "indrop_v3"][it % 2]
    n = int(rng.integers(150_000, 1_500_000))
    kw = dict(n_cells=int(rng.integers(10, 200)), n_genes=int(rng.integers(300, 12000)), umi_len=int(rng.integers(6, 13)),
              permille_neighbour=int(rng.integers(30, 250)), stream_id=int(rng.integers(1, 1000)))
    poisson = it % 3 == 2
    mb, ma, frac = int(rng.integers(1, 12)), int(rng.integers(5, 60)), [0.2, 0.0, 0.35][it % 3]
    s = SynthStream(n_reads=n, whitelist=wl, **kw)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    path = os.path.join(DATA, wl)
    t0 = time.time()
    o = parity.oracle_run(Oracle, dict(merge_kind=3 if poisson else 1, barcodes_kind=capi.BARCODES_CONST, barcodes_file=path, min_genes_before=mb,
                                       min_genes_after=ma, min_merge_fraction=frac), cb, umi, gene, aux)
    c = parity.gpu_run(dict(merge_kind=capi.MERGE_POISSON_REAL if poisson else capi.MERGE_REAL_BARCODES, barcodes_kind=capi.BARCODES_CONST,
                            barcodes_file=path, min_genes_before_merge=mb, min_genes_after_merge=ma, min_merge_fraction=frac),
                       cb, umi, gene, aux)
    parity.compare(o, c)
    mt = c.merge_targets()
    print(it, wl, n, kw, "poisson" if poisson else "-m", "merged", int((mt != np.arange(len(mt))).sum()),
          "excluded", int(c.cell_rows()["is_excluded"].sum()), "ok %.1fs" % (time.time() - t0), flush=True)
