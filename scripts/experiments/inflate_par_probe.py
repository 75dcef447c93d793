"""Probe of the lane-parallel inflate kernel (csrc/k_inflate_par.h, DROPEST_INFLATE_PAR=1): a few BGZF blocks of growing size and kind,
status and equality per block.  Run under a short `timeout`: a kernel that hangs must not cost the box."""
import os, sys, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_gpu_bgzf as t

rng = np.random.default_rng(1)
cases = [("tiny", b"hello world, hello world, hello"), ("1k_random", rng.integers(0, 256, 1000, dtype=np.uint8).tobytes()),
         ("acgt_8k", rng.choice(np.frombuffer(b"ACGT\n", np.uint8), 8000).tobytes()), ("text_64k", (b"the quick brown fox jumps over the lazy dog; " * 1500)[:64000]),
         ("runs", b"".join(bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 700)) for _ in range(180))[:65000])]
only = os.environ.get("PROBE_ONLY")
for name, data in cases:
    if only and name != only:
        continue
    for level in ((6,) if only else (1, 6, 9)):
        blob = t.bgzf_block(data, level)
        out, status, ms = t.inflate(blob)
        print(name, level, "status", list(status), "equal", out == data, "ms %.3f" % ms, flush=True)
