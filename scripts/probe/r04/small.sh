#!/bin/bash
cd /root/repo
export DROPEST_BENCH_NO_FORMS=1
B="python bench.py --no-secondary --steps 16 --warmup 3 --cpu-sample 0 --push-sample 0"
for i in 1 2; do
$B 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']; s=sorted(d['step_ms'])
print('c2', d['ms_per_step'], s[8], 'ksum', d['roofline']['pipeline']['kernel_ms_per_step'], {n: k[n]['ms_per_step'] for n in ('cb_sample','cb_hot_count','cb_hot_collect','cb_compact_slots','cb_assign_ids','ingest_sample_stats') if n in k})"
done
