import ctypes as C, time, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from dropest_amd import capi
import test_merge_apply_cpu as t
rng = np.random.default_rng(5)
n = 2_400_000
is_target = np.zeros(n, bool); is_target[rng.choice(n, 50000, replace=False)] = True
tg_ids = np.flatnonzero(is_target)
order = rng.permutation(n).astype(np.uint32)
target = np.where(is_target[order], order, tg_ids[rng.integers(0, len(tg_ids), n)]).astype(np.int64)
reads = rng.integers(1, 1000, n).astype(np.int32); umis = rng.integers(1, 500, n).astype(np.int32)
for rep in range(4):
    t0 = time.time(); out = t._apply(n, order, target, reads, umis); dt = time.time() - t0
    print("apply %.1f ms merged %d" % (dt * 1e3, int((out[2] != np.arange(n)).sum())))
