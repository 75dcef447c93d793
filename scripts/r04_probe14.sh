#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
T=${1:-r04w}
mkdir -p gpurun_out/$T
python -m pytest tests/test_gpu_ssort.py tests/test_gpu_stress.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/$T/tests.log 2>&1; echo "tests rc $?"; grep -E "passed|failed" gpurun_out/$T/tests.log | tail -2
B="python bench.py --no-secondary --steps 30 --warmup 3 --cpu-sample 0 --push-sample 0"
export DROPEST_BENCH_NO_FORMS=1
for i in 1 2; do
$B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=sorted(d['step_ms']); print('early', d['ms_per_step'], s[15], 'ksum', d['roofline']['pipeline']['kernel_ms_per_step'])"
DROPEST_SS_NO_EARLY_SAMPLE=1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=sorted(d['step_ms']); print('late ', d['ms_per_step'], s[15], 'ksum', d['roofline']['pipeline']['kernel_ms_per_step'])"
done
$B --config c3 --reads 1e9 --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', d['ms_per_step'], d['step_ms'])"
