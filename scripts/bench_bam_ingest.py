"""Developer tool: throughput of the native BAM ingest (BGZF inflate on a thread pool + record / tag parsing +
CellsDataContainer::add_record) on a synthetic 10x-style BAM, next to the estimation and output stages."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                   # noqa: E402
import bam_writer as bw                              # noqa: E402
from dropest_amd import capi                         # noqa: E402
from dropest_amd.build import build_facade           # noqa: E402
from dropest_amd.synth import SynthStream            # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
build_facade()
s = SynthStream(n_reads=n, n_cells=500, n_genes=5000, umi_len=10)
cb, umi, gene, aux = s.generate_host()
t0 = time.time()
refs = [("chr%d" % i, 10_000_000) for i in range(25)]
cbs = [capi.unpack_code(c) for c in np.unique(cb)]
cb_of = dict(zip(np.unique(cb).tolist(), cbs))
recs = []
for i in range(n):
    tags = [("CB", "Z", cb_of[int(cb[i])]), ("UB", "Z", capi.unpack_code(umi[i]))]
    if os.environ.get("QUAL"):      # UQ tags as a 10x BAM carries them: the readers keep one quality row per read
        tags.append(("UQ", "Z", "FFFFFFFFFF"))
    if gene[i] != capi.NO_GENE:
        tags.append(("GX", "Z", "ENSG%011d" % gene[i]))
    recs.append(bw.record(int(aux[i]) & 0xFFFF, i, "A00000:1:HXXXX:1:1101:%d:%d" % (i, i), seq="ACGT" * 24 + "AC", tags=tags))
if os.environ.get("REAL"):      # bases drawn at random and binned qualities: the file deflates ~3.2 x, as real 10x BAMs do (not 10.8 x)
    rng = np.random.default_rng(5)
    nib = rng.choice(np.array([1, 2, 4, 8], np.uint8), (n, 98))
    packed_seq = ((nib[:, 0::2] << 4) | nib[:, 1::2]).astype(np.uint8)
    quals = rng.choice(np.array([37, 25, 11, 2], np.uint8), (n, 98), p=[0.75, 0.12, 0.08, 0.05])
    for i in range(n):
        r = bytearray(recs[i])
        o = 36 + r[12] + 4
        r[o:o + 49] = packed_seq[i].tobytes(); r[o + 49:o + 147] = quals[i].tobytes()
        recs[i] = bytes(r)
tmp = tempfile.mkdtemp()
bam = os.path.join(tmp, "synth.bam")
copies = int(os.environ.get("COPIES", "16"))
bw.write_bam(bam, refs, recs, repeat=copies)
print("wrote %s: %d reads, %.1f MB, %.1f s" % (bam, n, os.path.getsize(bam) / 1e6, time.time() - t0), file=sys.stderr)
# (the records are written `copies` times into ONE file -- bam_writer.write_bam(repeat=...) -- so that a rate of tens of Mreads/s
# is measured over seconds, and inflate / parse / push pipeline as they do on a real multi-gigabyte BAM)
thread_counts = [int(x) for x in os.environ.get("THREADS", "1,4,16,64").split(",")]
for threads in thread_counts:
    for env, label in (({}, "bulk"), ({"DROPEST_BAM_RECORD_BY_RECORD": "1"}, "record-by-record"), ({"DROPEST_BAM_DEVICE": "1", "DROPEST_BAM_TRACE": "1"}, "device")):
        if label == "record-by-record" and threads not in (16,):
            continue
        if label == "device" and threads not in (4, 16):
            continue
        res = subprocess.run([os.path.join(ROOT, "tests", "cpp", "bam_to_counts"), os.path.join(tmp, "out"), "filled", "20", "100", "-",
                              str(threads), bam], capture_output=True, text=True, env=dict(os.environ, **env))
        if res.returncode:
            raise SystemExit(res.stderr)
        for line in res.stderr.splitlines():
            if line.startswith("[bam]"):
                print(line, file=sys.stderr)
        st = json.loads(res.stdout.strip().splitlines()[-1])
        st.update(threads=threads, path=label, reads=n * copies, ingest_mreads_per_s=round(n * copies / st["ingest_ms"] / 1e3, 3), bam_mb=round(os.path.getsize(bam) / 1e6, 1))
        print(json.dumps(st), flush=True)
if os.environ.get("KEEP_BAM"):
    import shutil; shutil.copy(bam, os.environ["KEEP_BAM"])
import shutil; shutil.rmtree(tmp, ignore_errors=True)
