"""Idle time of the device inside one pass: from a rocprofv3 --kernel-trace CSV (kernel start / end timestamps) of a bench run, the gaps
between consecutive kernels of one pass (TRACE_PASS: 0-based from the start, negative from the end, default the last one; bench.py --steps 2 --warmup 1
runs the warm-up, two timed steps, then table steps with HIP events around every launch and the other matrix forms: 2 is the second timed step), largest first, with the
kernels on either side.  usage: trace_gaps.py <dir with *kernel_trace.csv> [how many of the pass's last kernels to list on a time line]"""
import csv, glob, os, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last pass: from the last cb_sample_distinct launch on
starts = [i for i, r in enumerate(rows) if "cb_sample_distinct" in r[2]]
which = int(os.environ.get("TRACE_PASS", "-1"))
lo = starts[which]
which = which % len(starts)
hi = starts[which + 1] if which + 1 < len(starts) else len(rows)
print("pass %d of %d in the trace (TRACE_PASS: 0-based from the start, negative from the end)" % (which + 1, len(starts)))
seg = rows[lo:hi]
busy = sum(e - s for s, e, _ in seg)
span = max(e for _, e, _ in seg) - seg[0][0]
print("kernels %d, span %.3f ms, busy %.3f ms, idle %.3f ms" % (len(seg), span / 1e6, busy / 1e6, (span - busy) / 1e6))
gaps = []
end = seg[0][1]
for i in range(1, len(seg)):
    s, e, n = seg[i]
    if s > end:
        gaps.append((s - end, seg[i - 1][2][:60], n[:60]))
    end = max(end, e)
gaps.sort(reverse=True)
for g, a, b in gaps[:25]:
    print("%8.1f us  after %-60s before %s" % (g / 1e3, a, b))
if len(sys.argv) > 2:   # the last kernels of the pass on a time line (ms before the end of the pass)
    t_end = max(e for _, e, _ in seg)
    for s_, e_, n_ in seg[-int(sys.argv[2]):]:
        print("  start -%7.3f ms  dur %7.3f ms  %s" % ((t_end - s_) / 1e6, (e_ - s_) / 1e6, n_[:70]))
print("gaps > 20 us: %d, their sum %.3f ms; gaps <= 20 us: sum %.3f ms" % (sum(1 for g in gaps if g[0] > 20000), sum(g[0] for g in gaps if g[0] > 20000) / 1e6,
                                                                        sum(g[0] for g in gaps if g[0] <= 20000) / 1e6))
