cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD
OUT=gpurun_out/kt_easy; rm -rf $OUT; mkdir -p $OUT
THREADS=16 COPIES=64 KEEP_BAM=$PWD/$OUT/easy.bam timeout 600 python scripts/bench_bam_ingest.py 250000 > /dev/null 2>&1
(cd /tmp && export TMPDIR=/tmp && DROPEST_BAM_DEVICE=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/kt -o r -- $GRAFT_REPO_ROOT/tests/cpp/bam_to_counts /tmp/res filled 20 100 - 16 $GRAFT_REPO_ROOT/$OUT/easy.bam > /dev/null 2>&1)
rm -f $OUT/easy.bam
python - <<'PY'
import csv, glob
k = glob.glob("gpurun_out/kt_easy/kt/**/*kernel_trace.csv", recursive=True)[0]
m = glob.glob("gpurun_out/kt_easy/kt/**/*memory_copy_trace.csv", recursive=True)
ev = []
for r in csv.DictReader(open(k)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:50], r.get("Queue_Id", "")))
if m:
    for r in csv.DictReader(open(m[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), ""))
ev.sort()
t0 = ev[0][0]
# the last inflate kernel and what follows it
last = max(i for i, e in enumerate(ev) if "inflate" in e[2])
for s, e, n, q in ev[last - 3:last + 60]:
    print("%9.3f %9.3f  %8.3f ms  q%s  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n))
PY
