"""N plain passes of the bench's C2 (or c3) workload and nothing else -- the command to put under rocprofv3 --kernel-trace for
scripts/trace_gaps.py (bench.py itself runs a kernel-table pass, the other matrix forms and the ingest legs after its timed region).
usage: plain_passes.py [config] [reads] [passes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dropest_amd import capi
from dropest_amd.synth import SynthStream

config = sys.argv[1] if len(sys.argv) > 1 else "c2"
reads = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 6
c3 = config == "c3"
wl = os.path.join(ROOT, "dropest_amd", "data", "barcodes", "10x_aug_2016_split")
stream = SynthStream(n_reads=reads, n_cells=50000 if c3 else 5000, n_genes=30000, cb_len=16, whitelist="10x_aug_2016_split",
                     umi_len=12 if c3 else 10, stream_id=3 if c3 else 2)
dev = stream.generate_device(0, first=0, n=reads)
kw = dict(merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=capi.BARCODES_CONST, barcodes_file=wl, min_merge_fraction=0.2) if c3 else dict(merge_kind=capi.MERGE_NONE)
ctx = capi.Context(device=0, min_genes_before_merge=20, min_genes_after_merge=100, **kw)
ctx.push_reads_device(*dev.ptrs, dev.n, adopt=True)
for i in range(passes):
    t0 = time.perf_counter()
    bench.one_step(ctx)
    print("pass %d: %.3f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
