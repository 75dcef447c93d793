#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
T=${1:-r04n}
mkdir -p gpurun_out/$T
lscpu | grep -i "numa\|socket\|model name" ; cat /sys/devices/system/node/node*/cpulist | head -8
B="python bench.py --no-secondary --steps 10 --warmup 3 --cpu-sample 0 --push-sample 0"
export DROPEST_BENCH_NO_FORMS=1
for i in 1 2; do
DROPEST_TAIL_TRACE=1 $B 2> gpurun_out/$T/tail_numa$i.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('numa', d['ms_per_step'], sorted(d['step_ms']))"
grep "matrix done\|stream drained" gpurun_out/$T/tail_numa$i.err | tail -4
DROPEST_DECODE_NUMA=0 DROPEST_TAIL_TRACE=1 $B 2> gpurun_out/$T/tail_nonuma$i.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('free', d['ms_per_step'], sorted(d['step_ms']))"
grep "matrix done\|stream drained" gpurun_out/$T/tail_nonuma$i.err | tail -4
done
