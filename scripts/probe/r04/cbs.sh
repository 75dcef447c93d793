#!/bin/bash
cd /root/repo
export DROPEST_BENCH_NO_FORMS=1
B="python bench.py --no-secondary --steps 16 --warmup 3 --cpu-sample 0 --push-sample 0"
for e in 1 4 8 1 4 8; do
DROPEST_CB_SAMPLE_COUNT_EVERY=$e $B 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']; s=sorted(d['step_ms'])
print('every $e', d['ms_per_step'], s[8], 'ksum', d['roofline']['pipeline']['kernel_ms_per_step'], 'cb_sample', k.get('cb_sample',{}).get('ms_per_step'), 'cb_insert', k.get('cb_insert',{}).get('ms_per_step'), 'build_keys', k.get('build_keys',{}).get('ms_per_step'))"
done
for e in 1 4; do
DROPEST_CB_SAMPLE_COUNT_EVERY=$e $B --config c3 --reads 1e9 --steps 4 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']; s=sorted(d['step_ms'])
print('c3 every $e', d['ms_per_step'], s, 'cb_sample', k.get('cb_sample',{}).get('ms_per_step'), 'cb_insert', k.get('cb_insert',{}).get('ms_per_step'), 'build_keys', k.get('build_keys',{}).get('ms_per_step'))"
done
