import os, sys, zlib
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import test_gpu_bgzf as t
src = open(os.path.join(ROOT, "scripts", "soak_inflate.py")).read()
ns = {"__file__": os.path.join(ROOT, "scripts", "soak_inflate.py")}
exec(src.split("bad = 0")[0].replace("rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20", "rounds = 0").replace("int(sys.argv[2]) if len(sys.argv) > 2 else 7", "7"), ns)
payload, STRATS = ns["payload"], ns["STRATS"]
names = ["noise", "few symbols", "runs", "periods", "bam-like", "far repeats", "text"]
for kind in range(7):
    for strat, sn in ((zlib.Z_DEFAULT_STRATEGY, "default"), (zlib.Z_HUFFMAN_ONLY, "huffman"), (zlib.Z_RLE, "rle"), (zlib.Z_FIXED, "fixed")):
        for level in (1, 6):
            datas = [payload(kind, 65000) for _ in range(64)]
            blob = b"".join(t.bgzf_block(d, level, strat) for d in datas)
            out, status, ms = t.inflate(blob, repeats=2)
            print("%-12s %-8s level %d: 64 blocks, %6.0f KB in, kernel %7.3f ms  %s" % (names[kind], sn, level, len(blob) / 1e3, ms, "ok" if out == b"".join(datas) and not status.any() else "WRONG"), flush=True)
