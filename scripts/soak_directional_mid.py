"""Soak (not part of the suites): random mid-size streams through -u (directional UMI correction), with and without N in
the UMIs, GPU against the oracle.  Run on a GPU box: PYTHONPATH=. python scripts/soak_directional_mid.py"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from dropest_amd import capi
from dropest_amd.synth import SynthStream, inject_n
from oracle import Oracle
import parity
libc = ctypes.CDLL("libc.so.6")
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "55")))
for it in range(int(os.environ.get("SOAK_CASES", "10"))):
    n = int(rng.integers(100_000, 900_000))
    umi_len = int(rng.integers(4, 9))
    kw = dict(n_cells=int(rng.integers(5, 120)), n_genes=int(rng.integers(40, 3000)), umi_len=umi_len, stream_id=int(rng.integers(1, 1000)))
    max_ed, mult, mg = int(rng.integers(1, 4)), [1.0, 1.5, 2.0, 3.0][int(rng.integers(0, 4))], int(rng.integers(1, 8))
    s = SynthStream(n_reads=n, **kw)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    side = ()
    if it % 2:
        umi, side = inject_n(umi, gene, 10 ** -float(rng.uniform(2, 3.5)), int(rng.integers(1, 1000)), umi_len)
    okw = dict(umi_merge_kind=1, max_umi_merge_ed=max_ed, umi_mult=mult, min_genes_before=mg, min_genes_after=mg)
    gkw = dict(umi_merge_kind=capi.UMI_MERGE_DIRECTIONAL, max_umi_merge_edit_distance=max_ed, umi_merge_multiplier=mult,
               min_genes_before_merge=mg, min_genes_after_merge=mg)
    t0 = time.time()
    libc.srand(1); o = parity.oracle_run(Oracle, okw, cb, umi, gene, aux, side)
    libc.srand(1); c = parity.gpu_run(gkw, cb, umi, gene, aux, side)
    parity.compare(o, c, side)
    print(it, n, kw, "ed", max_ed, "mult", mult, "N" if it % 2 else "-", "molecules", int(c.molecules()[0].shape[0]), "ok %.1fs" % (time.time() - t0), flush=True)
