#!/bin/bash
cd /root/repo
export DROPEST_BENCH_NO_FORMS=1
B="python bench.py --no-secondary --cpu-sample 0 --push-sample 0"
for m in 200000 100000 60000; do
DROPEST_DEVICE_SORT_MIN=$m $B --config c4 --reads 1.25e8 --steps 6 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_stage_wall_ms_per_step']
print('c4 min $m', d['ms_per_step'], sorted(d['step_ms']), {k:v for k,v in h.items() if k.startswith('sort_filtered')})"
done
for m in 200000 100000; do
DROPEST_DEVICE_SORT_MIN=$m $B --config c3 --reads 1e9 --steps 4 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_stage_wall_ms_per_step']
print('c3 min $m', d['ms_per_step'], sorted(d['step_ms']), {k:v for k,v in h.items() if k.startswith('sort_filtered')})"
done
