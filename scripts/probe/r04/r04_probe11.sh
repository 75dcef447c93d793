#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
T=${1:-r04p}
mkdir -p gpurun_out/$T
B="python bench.py --no-secondary --steps 30 --warmup 3 --cpu-sample 0 --push-sample 0"
export DROPEST_BENCH_NO_FORMS=1
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -3; rocm-smi --showtoponuma 2>/dev/null | grep -i numa | head -4
for i in 1 2 3 4; do
DROPEST_WIRE_TRACE=1 $B 2> gpurun_out/$T/t$i.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=sorted(d['step_ms']); print('wire', d['ms_per_step'], s[15], s[-3:])"
grep "nodes:" gpurun_out/$T/t$i.err | sort | uniq -c | sort -rn | head -4
done
