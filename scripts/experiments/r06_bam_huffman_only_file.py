# a BAM whose blocks are Huffman-only streams of few symbols: the decoder should switch kernels after two slow windows, and the container is the host reader's
import os, sys, subprocess, zlib, json, struct
R = os.environ["GRAFT_REPO_ROOT"]; sys.path.insert(0, R + "/tests"); sys.path.insert(0, R)
import numpy as np
import bam_writer as bw
rng = np.random.default_rng(3)
refs = [("chr1", 1000000)]
recs = []
# long reads of N with qualities drawn from six values: nearly every byte of the file is one of seven, Huffman-only codes of 2-3 bits that never fall in step
for i in range(12000):
    cb = "".join(rng.choice(list("ACGT"), 16))
    rec = bytearray(bw.record(0, i, "r", seq="N" * 8000, tags=[("CB", "Z", cb), ("UB", "Z", "ACGTACGTAC"), ("GX", "Z", "G%d" % (i % 50))]))
    o = 36 + 2 + 4 + 4000                                   # block_size + fixed fields + the name "r\0" + one CIGAR operation + 4 000 bytes of packed bases
    rec[o:o + 8000] = rng.choice(np.array([2, 11, 25, 37, 40, 41], np.uint8), 8000).tobytes()
    recs.append(bytes(rec))
# blocks written with Z_HUFFMAN_ONLY
orig = bw._bgzf_block
def hblock(data):
    c = zlib.compressobj(6, zlib.DEFLATED, -15, 9, zlib.Z_HUFFMAN_ONLY); payload = c.compress(data) + c.flush()
    bsize = len(payload) + 26
    return struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize - 1) + payload + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))
bw._bgzf_block = hblock
bw.write_bam("/tmp/slow.bam", refs, recs, block=60000)
print("file MB", os.path.getsize("/tmp/slow.bam") / 1e6)
out = {}
for name, env in (("host", {}), ("device", {"DROPEST_BAM_DEVICE": "1", "DROPEST_BAM_TRACE": "1"}), ("device_windows_of_8MB", {"DROPEST_BAM_DEVICE": "1", "DROPEST_BAM_TRACE": "1", "DROPEST_BAM_DEVICE_WINDOW_MB": "8"}), ("device_par_only", {"DROPEST_BAM_DEVICE": "1", "DROPEST_INFLATE_PAR": "1"}), ("device_serial", {"DROPEST_BAM_DEVICE": "1", "DROPEST_INFLATE_PAR": "0"})):
    r = subprocess.run([R + "/tests/cpp/bam_to_counts", "/tmp/res_" + name, "filled", "1", "1", "-", "8", "/tmp/slow.bam"], env=dict(os.environ, **env), capture_output=True, text=True)
    j = json.loads(r.stdout.strip().splitlines()[-1]); out[name] = j
    print(name, j["ingest_ms"], j["saved"], j["cells"], [l for l in r.stderr.splitlines() if "per round" in l or "windows," in l][:2])
