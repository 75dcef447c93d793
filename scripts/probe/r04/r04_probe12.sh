#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
T=${1:-r04q}
mkdir -p gpurun_out/$T
python -m pytest tests/test_gpu_wire.py -x -q 2>&1 | tail -2
B="python bench.py --no-secondary --steps 60 --warmup 3 --cpu-sample 0 --push-sample 0"
export DROPEST_BENCH_NO_FORMS=1
rocm-smi --showtoponuma 2>/dev/null | grep -i "numa node" | head -2
for i in 1 2 3 4; do $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=sorted(d['step_ms']); print('wire numa', d['ms_per_step'], s[30], s[-4:])"; done
for i in 1 2; do DROPEST_DECODE_NUMA=0 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=sorted(d['step_ms']); print('wire free', d['ms_per_step'], s[30], s[-4:])"; done
DROPEST_BENCH_MATRIX_FORM=bytes $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=sorted(d['step_ms']); print('bytes', d['ms_per_step'], s[30], s[-4:])"
