#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
export DROPEST_BENCH_NO_FORMS=1
timeout 1500 python -m pytest tests/test_gpu_ssort.py -x -q > gpurun_out/os3_tests.log 2>&1; echo "ssort rc $?"; tail -1 gpurun_out/os3_tests.log
B="python bench.py --no-secondary --cpu-sample 0 --push-sample 0"
run() { $B $2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']; s=sorted(d['step_ms'])
print('$1', d['ms_per_step'], s[len(s)//2], 'ksum', d['roofline']['pipeline']['kernel_ms_per_step'], {n: round(k[n]['ms_per_step'],2) for n in k if n.startswith('ss_')})"; }
run c3 "--config c3 --reads 1e9 --steps 4 --warmup 1"
DROPEST_SSORT_OS=64 run c3_os64 "--config c3 --reads 1e9 --steps 4 --warmup 1"
run c2_6e7 "--reads 6e7 --steps 10 --warmup 3"
DROPEST_SSORT_OS=64 run c2_6e7_os64 "--reads 6e7 --steps 10 --warmup 3"
run c2_1e8 "--steps 10 --warmup 3"
