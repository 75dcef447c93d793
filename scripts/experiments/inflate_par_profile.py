"""Where a block's time goes in the lane-parallel inflate (csrc/k_inflate_par.h built with -DINFP_PROFILE: scripts/experiments/inflate_variants/
libbgzf_par_prof.so), on blocks of the file that deflates 3.2 x, for a growing number of blocks per launch.
usage: DROPEST_BGZF_LIB=.../libbgzf_par_prof.so python scripts/experiments/inflate_par_profile.py [reads]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bam_writer as bw
import test_gpu_bgzf as tb
from dropest_amd import capi
from dropest_amd.synth import SynthStream

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000
real = os.environ.get("REAL", "1") != "0"
s = SynthStream(n_reads=n, n_cells=500, n_genes=5000, umi_len=10)
cb, umi, gene, aux = s.generate_host()
cbs = {int(c): capi.unpack_code(c) for c in np.unique(cb)}
rng = np.random.default_rng(5)
nib = rng.choice(np.array([1, 2, 4, 8], np.uint8), (n, 98))
packed_seq = (nib[:, 0::2] << 4) | nib[:, 1::2]
quals = rng.choice(np.array([37, 25, 11, 2], np.uint8), (n, 98), p=[0.75, 0.12, 0.08, 0.05])
body = bytearray()
for i in range(n):
    tags = [("CB", "Z", cbs[int(cb[i])]), ("UB", "Z", capi.unpack_code(umi[i]))]
    if gene[i] != capi.NO_GENE:
        tags.append(("GX", "Z", "ENSG%011d" % gene[i]))
    name = "A00000:1:HXXXX:1:1101:%d:%d" % (i, i)
    rec = bytearray(bw.record(int(aux[i]) & 0xFFFF, i, name, seq="ACGT" * 24 + "AC", tags=tags))
    if real:
        o = 36 + len(name) + 1 + 4
        rec[o:o + 49] = packed_seq[i].tobytes(); rec[o + 49:o + 147] = quals[i].tobytes()
    body += rec
body = bytes(body)
blocks = [bw._bgzf_block(body[o:o + 0xFF00]) for o in range(0, len(body), 0xFF00)][:-1]
print("blocks made:", len(blocks), "mean compressed bytes", sum(map(len, blocks)) // len(blocks), flush=True)
L = C.CDLL(os.environ["DROPEST_BGZF_LIB"])
prof = (C.c_ulonglong * 24)()
have_prof = hasattr(L, "dropest_bgzf_inflate_profile")
names = ["header", "span_in", "A", "B", "C", "D", "crc"]
for nb in (1, 1024, 4096, 16384):
    blob = b"".join((blocks * (nb // len(blocks) + 1))[:nb])
    if have_prof: L.dropest_bgzf_inflate_profile(prof)
    out, status, ms = tb.inflate(blob, repeats=3)
    line = "blocks %6d  kernel %.3f ms  %.1f GB/s  ok %s" % (nb, ms, len(out) / ms / 1e6, not status.any())
    if have_prof:
        L.dropest_bgzf_inflate_profile(prof)
        p = [int(x) for x in prof]
        k = max(1, p[11])
        line += " | per block us: " + " ".join("%s %.0f" % (names[j], p[j] / k / 100.0) for j in range(7))
        line += " | D: window %.0f rounds %.0f flush %.0f us; spans/block %.1f, A rounds/span %.1f, D batches/block %.0f, D rounds/block %.0f, matches/block %.0f, serial fall-backs/block %.2f" % (
            p[9] / k / 100.0, p[10] / k / 100.0, p[13] / k / 100.0, p[8] / k, p[7] / max(1, p[8]), p[15] / k, p[14] / k, p[12] / k, p[16] / k)
        line += ", walking rounds of (A)/span %.1f" % (p[17] / max(1, p[8]))
    print(line, flush=True)
