#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
export DROPEST_BENCH_NO_FORMS=1
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_gpu_multi.py tests/test_golden.py tests/test_gpu_quality.py -x -q -k "merge or whitelist or 1e9 or full_size or c4 or golden or qualit" > gpurun_out/apply_tests.log 2>&1; echo "rc $?"; tail -1 gpurun_out/apply_tests.log
B="python bench.py --no-secondary --cpu-sample 0 --push-sample 0"
for i in 1 2; do
$B --config c3 --reads 1e9 --steps 4 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_stage_wall_ms_per_step']; s=sorted(d['step_ms'])
print('c3', d['ms_per_step'], s, 'ksum', d['roofline']['pipeline']['kernel_ms_per_step'], {k: round(v,2) for k,v in h.items() if k.startswith('cb_merge:re') or k=='cb_merge'})"
done
$B --config c4 --reads 1.25e8 --steps 6 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=sorted(d['step_ms']); print('c4', d['ms_per_step'], s)"
