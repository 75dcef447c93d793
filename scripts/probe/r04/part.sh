#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_multi.py tests/test_gpu_multiproc.py -x -q > gpurun_out/part_tests.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/part_tests.log
export DROPEST_BENCH_NO_FORMS=1
B="python bench.py --no-secondary --steps 20 --warmup 3 --cpu-sample 0 --push-sample 0"
for i in 1 2; do
$B --sharded 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_stage_wall_ms_per_step']
print('sharded sampled', d['ms_per_step'], sorted(d['step_ms'])[10], h.get('shard:partition'))"
done
DROPEST_BENCH_MATRIX_FORM=bytes $B 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain bytes', d['ms_per_step'], sorted(d['step_ms'])[10])"
$B 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain u32', d['ms_per_step'], sorted(d['step_ms'])[10])"
