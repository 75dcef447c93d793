#!/bin/bash
# full GPU suite + the driver's bench command; results under gpurun_out/$1
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
T=${1:-r04b}
mkdir -p gpurun_out/$T
python -m pytest tests -x -q -m gpu > gpurun_out/$T/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/$T/tests.log
tail -4 gpurun_out/$T/tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$T/bench_default.json 2> gpurun_out/$T/bench_default.err
python - $T <<'P'
import json,sys
d=json.loads(open('gpurun_out/%s/bench_default.json'%sys.argv[1]).read().strip().splitlines()[-1])
print('C2', d['ms_per_step'], d['value'], d['config']['matrix_forms'])
print('roof', {k:d['roofline'][k] for k in ('kernel','frac','avg_launch_ms')}, d['roofline']['pipeline']['kernel_ms_per_step'])
s=d.get('secondary',{})
for k,v in s.items():
    print(k, v.get('ms_per_step'), v.get('step_ms'), v.get('x_plain'), v.get('error'))
print({k:v for k,v in d['host_stage_wall_ms_per_step'].items()})
P
