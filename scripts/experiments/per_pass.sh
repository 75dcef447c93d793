#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o r -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(list(rows[0].keys()))
gk = [k for k in rows[0] if "grid" in k.lower()]
def grid(r):
    return int(r[gk[0]]) * (int(r[gk[1]]) if len(gk) > 1 else 1) if gk else 0
big = [r for r in rows if ("rs_scatter" in r["Kernel_Name"] or "rs_hist" in r["Kernel_Name"])]
big = [r for r in big if (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) > 60000]
for r in big[-16:]:
    print(("scatter" if "scatter" in r["Kernel_Name"] else "hist   "), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
