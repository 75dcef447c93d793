"""Developer tool: rate of the device BGZF inflate (csrc/k_inflate_par.h; DROPEST_INFLATE_PAR=0: csrc/k_inflate.h) on a synthetic 10x-style BAM: GB/s of inflated bytes and the
Mreads/s that corresponds to, beside zlib on one host thread.  usage: python scripts/bench_bgzf_inflate.py [reads] [copies]"""
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                   # noqa: E402
import bam_writer as bw                              # noqa: E402
import test_gpu_bgzf as tb                           # noqa: E402
from dropest_amd import capi                         # noqa: E402
from dropest_amd.synth import SynthStream            # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 300_000
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 40
s = SynthStream(n_reads=n, n_cells=500, n_genes=5000, umi_len=10)
cb, umi, gene, aux = s.generate_host()
cbs = {int(c): capi.unpack_code(c) for c in np.unique(cb)}
body = bytearray()
# REAL=1: bases drawn at random and binned qualities as a NovaSeq writes them (75 % 'F', the rest ':' ',' '#') instead of one sequence and no
# qualities: the file inflates 3-4 x, as real 10x BAMs do, not 10.8 x -- most symbols are literals then
real = bool(os.environ.get("REAL"))
if real:
    rng = np.random.default_rng(5)
    nib = rng.choice(np.array([1, 2, 4, 8], np.uint8), (n, 98))
    packed_seq = (nib[:, 0::2] << 4) | nib[:, 1::2]
    quals = rng.choice(np.array([37, 25, 11, 2], np.uint8), (n, 98), p=[0.75, 0.12, 0.08, 0.05])
for i in range(n):
    tags = [("CB", "Z", cbs[int(cb[i])]), ("UB", "Z", capi.unpack_code(umi[i]))]
    if gene[i] != capi.NO_GENE:
        tags.append(("GX", "Z", "ENSG%011d" % gene[i]))
    name = "A00000:1:HXXXX:1:1101:%d:%d" % (i, i)
    rec = bw.record(int(aux[i]) & 0xFFFF, i, name, seq="ACGT" * 24 + "AC", tags=tags)
    if real:
        rec = bytearray(rec)
        o = 36 + len(name) + 1 + 4          # block_size + the fixed fields + the name + one CIGAR operation
        rec[o:o + 49] = packed_seq[i].tobytes(); rec[o + 49:o + 147] = quals[i].tobytes()
    body += rec
body = bytes(body)
blocks = [bw._bgzf_block(body[o:o + 0xFF00]) for o in range(0, len(body), 0xFF00)]
blob = b"".join(blocks) * copies
t0 = time.time()
d = zlib.decompressobj(31)
host_bytes = 0
for b in blocks:
    host_bytes += len(zlib.decompress(b[18:-8], -15))
host_s = time.time() - t0
out, status, ms = tb.inflate(blob, repeats=5)
_, _, ms_nocrc = tb.inflate(blob, repeats=-5)      # the same without the CRC-32 of every block
ok = not status.any() and out[:len(body)] == body and out[-len(body):] == body
gb = len(out) / 1e9
print(json.dumps({"data": "random bases, binned qualities" if real else "one sequence, no qualities", "ratio": round(len(out) / len(blob), 2), "reads": n * copies, "blocks": len(status), "inflated_GB": round(gb, 3), "compressed_GB": round(len(blob) / 1e9, 3), "bytes_per_read": round(len(body) / n, 1),
                  "kernel_ms": round(ms, 3), "kernel_ms_without_crc32": round(ms_nocrc, 3), "device_GB_per_s": round(gb / ms * 1e3, 2), "device_Mreads_per_s": round(n * copies / ms / 1e3, 1),
                  "zlib_one_thread_GB_per_s": round(host_bytes / 1e9 / host_s, 3), "refused_blocks": int((status != 0).sum()), "identical": bool(ok)}))
