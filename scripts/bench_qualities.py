"""Cost of the UMI quality sums at C2 scale: python scripts/bench_qualities.py [n_reads]"""
import sys
import time

import numpy as np

from dropest_amd import capi
from dropest_amd.synth import SynthStream

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
s = SynthStream(n_reads=n, n_cells=5000, n_genes=30000, cb_len=16, umi_len=10, stream_id=2)
for with_q in (False, True):
    dev = s.generate_device(0, first=0, n=n)
    c = capi.Context(min_genes_before_merge=10, min_genes_after_merge=10)
    c.set_profiling(True)
    c.push_reads_device(*dev.ptrs, dev.n, adopt=True)
    if with_q:
        q = np.random.default_rng(1).integers(33, 75, size=(n, 10), dtype=np.uint8)
        t0 = time.time(); c.set_umi_qualities(q); print("set_umi_qualities (H2D of %.1f GB): %.1f ms" % (q.nbytes / 1e9, (time.time() - t0) * 1e3))
    t0 = time.time()
    c.set_initialized(); c.merge_and_filter()
    print("qualities=%s: set_initialized + merge_and_filter %.1f ms" % (with_q, (time.time() - t0) * 1e3))
    st = c.kernel_stats()
    for k, v in st.items():
        if "quality" in k:
            print(" ", k, v)
    del c, dev
