#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
export DROPEST_BENCH_NO_FORMS=1
B="python bench.py --no-secondary --cpu-sample 0 --push-sample 0 --config c3 --reads 1e9 --steps 3 --warmup 1"
run() { $B 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']; s=sorted(d['step_ms'])
print('$1', d['ms_per_step'], 'ksum', d['roofline']['pipeline']['kernel_ms_per_step'], {n: round(k[n]['ms_per_step'],2) for n in k if n.startswith('ss_')})"; }
DROPEST_SSORT_OS=32 DROPEST_SS_CAP2_PERCENT=300 run os32_cap300
DROPEST_SSORT_OS=48 DROPEST_SS_CAP2_PERCENT=200 run os48_cap200
DROPEST_SS_CAP2_PERCENT=200 run os64_cap200
