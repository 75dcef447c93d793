import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from dropest_amd import capi
from dropest_amd.multi import cfg_kwargs
from dropest_amd.synth import SynthStream, inject_n
import parity, test_gpu_multi as tm
SC = int(os.environ.get("SC", "12")); world = int(os.environ.get("W", "2")); rate = float(os.environ.get("RATE", "0.02"))
s = SynthStream(n_reads=150_000 * SC, n_cells=30 * SC, n_genes=400, umi_len=6, reads_per_molecule=3)
cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
umi, side = inject_n(umi, gene, rate, 13, 6)
kw = dict(cfg_kwargs({"min_before": 5, "min_after": 10}), umi_merge_kind=capi.UMI_MERGE_DIRECTIONAL, max_umi_merge_edit_distance=1, umi_merge_multiplier=2.0)
got = tm.run_group(world, (cb, umi, gene, aux), kw, side, steps=1)
c = tm.single((cb, umi, gene, aux), kw, side)
print({k: v for k, v in c.kernel_stats().items() if k.startswith("count:")})
for filt, name in ((True, "cm"), (False, "raw")):
    p, i, x = c.count_matrix_csc(filtered=filt)
    gp, gi, gx, gb = got[name]
    print(name, "ncols", len(p) - 1, len(gp) - 1, "nnz", len(i), len(gi), "sum", int(x.sum()), int(gx.sum()))
    if len(p) == len(gp):
        dc = np.flatnonzero(np.diff(p.astype(np.int64)) != np.diff(gp.astype(np.int64)))
        print("  columns with different nnz:", len(dc), dc[:10])
        if len(i) == len(gi):
            bad = np.flatnonzero((i != gi) | (x != gx)); print("  differing entries", len(bad), bad[:10])
            cols = np.searchsorted(p.astype(np.int64), bad, side="right") - 1
            print("  in columns", np.unique(cols)[:20])
