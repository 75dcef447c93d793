import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import parity
from dropest_amd.synth import SynthStream
import test_gpu_multi_oracle as t
arrays = parity.canonical_stream(*SynthStream(n_reads=200_000, n_cells=50, n_genes=30000).generate_host())
kw = dict(min_genes_before_merge=10, min_genes_after_merge=30)
bounds = t.even_bounds(len(arrays[0]), 3)
want = t.run_shards(arrays, kw, bounds)
os.environ["DROPEST_MATRIX_ROW_LIST_CAP"] = "12"
got = t.run_shards(arrays, kw, bounds, steps=1)
for k in ("cm", "raw"):
    for name, a, b in zip(("colptr", "rows", "vals", "bc"), got[k], want[k]):
        bad = np.flatnonzero(a != b)
        print(k, name, len(a), "mismatches", len(bad), bad[:10], a[bad[:5]], b[bad[:5]])
    p = want[k][0].astype(np.int64)
    bad = np.flatnonzero(got[k][1] != want[k][1])
    if len(bad):
        cols = np.searchsorted(p, bad, side="right") - 1
        print("  bad columns", np.unique(cols)[:20], "of", len(p) - 1)
print(got["phases"].get("matrix:overflow"))
