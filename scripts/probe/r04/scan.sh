#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
export DROPEST_BENCH_NO_FORMS=1
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ssort.py tests/test_gpu_stress.py -x -q > gpurun_out/scan_tests.log 2>&1; echo "rc $?"; tail -1 gpurun_out/scan_tests.log
B="python bench.py --no-secondary --cpu-sample 0 --push-sample 0"
for i in 1 2; do
$B --config c3 --reads 1e9 --steps 4 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']; s=sorted(d['step_ms'])
print('c3', d['ms_per_step'], s, 'ksum', d['roofline']['pipeline']['kernel_ms_per_step'], {n: round(k[n]['ms_per_step'],2) for n in k if 'scan' in n})"
done
