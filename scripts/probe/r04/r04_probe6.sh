#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
T=${1:-r04h}
mkdir -p gpurun_out/$T
B="python bench.py --no-secondary --steps 60 --warmup 3 --cpu-sample 0 --push-sample 0"
export DROPEST_BENCH_NO_FORMS=1
DROPEST_WIRE_TRACE=1 $B > gpurun_out/$T/bench_trace.json 2> gpurun_out/$T/trace.err
python - $T <<'P'
import json,sys,re
d=json.loads(open('gpurun_out/%s/bench_trace.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(d['ms_per_step'], sorted(d['step_ms'])[30], [x for x in d['step_ms'] if x>10.5])
rows=[l for l in open('gpurun_out/%s/trace.err'%sys.argv[1]) if l.startswith('[wire]')]
big=[l for l in rows if float(re.search(r'slowest slice ([0-9.]+)',l).group(1))>0.5 or float(re.search(r'done ([0-9.]+)',l).group(1))>4]
print(len(rows)); print(''.join(rows[10:14])); print('OUTLIERS'); print(''.join(big[:12]))
P
