#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
export DROPEST_BENCH_NO_FORMS=1
B="python bench.py --no-secondary --steps 12 --warmup 3 --cpu-sample 0 --push-sample 0"
run() { $B 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']; s=sorted(d['step_ms'])
print('$1', d['ms_per_step'], s[6], 'ksum', d['roofline']['pipeline']['kernel_ms_per_step'], {n: round(k[n]['ms_per_step'],3) for n in k if n.startswith('ss_')})"; }
run base
DROPEST_SSORT_TB=17 run tb17
DROPEST_SSORT_TB=17 DROPEST_SS_SMALL_MAX=1024 run tb17_cap1024
DROPEST_SSORT_TB=17 DROPEST_SS_SMALL_MAX=1024 DROPEST_SS_LOCAL_WAVE=128 run tb17_cap1024_w128
DROPEST_SSORT_TB=17 DROPEST_SS_SMALL_MAX=1024 DROPEST_SS_LOCAL_WAVE=64 run tb17_cap1024_w64
run base
