#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_ssort.py tests/test_gpu_stress.py -x -q > gpurun_out/ssl.log 2>&1; echo "rc $?"; tail -2 gpurun_out/ssl.log
export DROPEST_BENCH_NO_FORMS=1
B="python bench.py --no-secondary --steps 16 --warmup 3 --cpu-sample 0 --push-sample 0"
for i in 1 2 3; do
$B 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=sorted(d['step_ms']); h=d['host_stage_wall_ms_per_step']
print('c2', d['ms_per_step'], s[8], s[0], 'ksum', d['roofline']['pipeline']['kernel_ms_per_step'], 'splitter', h.get('splitter_sort'), 'decode_wait', h.get('matrix:decode_wait'))"
done
DROPEST_BENCH_MATRIX_FORM=bytes $B 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=sorted(d['step_ms']); print('bytes', d['ms_per_step'], s[8], s[0])"
