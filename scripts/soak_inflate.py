"""Soak of the device BGZF inflate (csrc/k_inflate_par.h by default; DROPEST_INFLATE_PAR=0: k_inflate.h) against zlib: rounds of a few hundred blocks of
random kinds -- bytes of a few symbols, text with far and near repeats, long runs, periods, BAM-like records, pure noise -- at random sizes, zlib levels
and strategies (default, filtered, Huffman only, RLE, fixed codes), some cut into several DEFLATE blocks; every round once more with a few payload bytes
flipped (a verdict per block, no byte outside a block's own range touched).  usage: python scripts/soak_inflate.py [rounds] [seed]"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_bgzf as t

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
STRATS = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]

def payload(kind, n):
    if kind == 0: return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == 1: return rng.choice(np.frombuffer(b"ACGTN\n", np.uint8), n).tobytes()
    if kind == 2:      # runs of every length, some longer than a match
        out = bytearray()
        while len(out) < n: out += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 1200))
        return bytes(out[:n])
    if kind == 3:      # short periods (distances 2..9) and their mixtures
        out = bytearray()
        while len(out) < n:
            p = rng.integers(0, 256, int(rng.integers(2, 10)), dtype=np.uint8).tobytes()
            out += p * int(rng.integers(1, 400))
        return bytes(out[:n])
    if kind == 4:      # records that repeat fields of the record before at ~300 bytes, with noise between (a BAM)
        out = bytearray(); prev = rng.integers(65, 91, 300, dtype=np.uint8).tobytes()
        while len(out) < n:
            cur = bytearray(prev)
            for _ in range(int(rng.integers(1, 12))):
                a = int(rng.integers(0, 290)); cur[a:a + int(rng.integers(1, 40))] = rng.integers(33, 127, int(rng.integers(1, 40)), dtype=np.uint8).tobytes()
            cur = bytes(cur[:int(rng.integers(200, 400))]); out += cur; prev = (cur * 2)[:300]
        return bytes(out[:n])
    if kind == 5:      # far repeats: the second half copies pieces of the first from up to 32 KB back
        half = rng.integers(0, 64, n // 2 + 1, dtype=np.uint8).tobytes(); out = bytearray(half)
        while len(out) < n:
            a = int(rng.integers(0, max(1, len(half) - 300))); out += half[a:a + int(rng.integers(3, 300))] or b"x"
        return bytes(out[:n])
    return (b"the quick brown fox jumps over the lazy dog; " * 1500)[:n]

bad = 0
for r in range(rounds):
    datas, blocks = [], []
    for _ in range(int(rng.integers(100, 400))):
        n = int(rng.choice([0, 1, 2, 3, 17, 300, 4000, 30000, 65000, 65280, int(rng.integers(1, 65281))]))
        d = payload(int(rng.integers(0, 7)), n) if n else b""
        datas.append(d)
        blocks.append(t.bgzf_block(d, int(rng.integers(0, 10)), STRATS[int(rng.integers(0, len(STRATS)))], flush_every=int(rng.choice([0, 0, 0, 5000, 20000]))))
    blob = b"".join(blocks)
    out, status, ms = t.inflate(blob)
    want = b"".join(datas)
    ok = not status.any() and out == want
    # the same with a few payload bytes flipped: every block gets a verdict, the good ones come out whole and in place
    hurt = bytearray(blob); offs = np.cumsum([0] + [len(b) for b in blocks]); victims = set()
    for _ in range(12):
        k = int(rng.integers(0, len(blocks)))
        if len(blocks[k]) > 40:
            hurt[int(offs[k]) + int(rng.integers(18, len(blocks[k]) - 8))] ^= 1 << int(rng.integers(0, 8)); victims.add(k)
    out2, status2, _ = t.inflate(bytes(hurt))
    oo = np.cumsum([0] + [len(d) for d in datas])
    ok2 = len(status2) == len(blocks) and all((status2[k] == 0 and out2[oo[k]:oo[k + 1]] == datas[k]) for k in range(len(blocks)) if k not in victims)
    ok2 = ok2 and all(status2[k] != 0 or out2[oo[k]:oo[k + 1]] == datas[k] for k in victims)      # (a flipped bit the CRC cannot miss; an accepted block is the right one)
    print("round %d: %d blocks, %.1f MB out, kernel %.2f ms: %s; %d blocks hurt: %s" % (r, len(blocks), len(want) / 1e6, ms, "SAME" if ok else "DIFFERENT", len(victims), "verdicts right" if ok2 else "WRONG"), flush=True)
    bad += (not ok) + (not ok2)
print("all the same" if not bad else "%d rounds differ" % bad)
sys.exit(1 if bad else 0)
