#!/bin/bash
cd /root/repo
export DROPEST_BENCH_NO_FORMS=1
B="python bench.py --no-secondary --steps 16 --warmup 3 --cpu-sample 0 --push-sample 0"
for rep in 1 2; do
for cfg in "14 65536" "20 32768" "28 32768" "28 16384" "40 16384" "14 32768"; do
set -- $cfg
DROPEST_DECODE_THREADS=$1 DROPEST_DECODE_SLICE=$2 $B 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_stage_wall_ms_per_step']; s=sorted(d['step_ms']); print('threads $1 slice $2', d['ms_per_step'], s[8], s[0], 'decode_wait', h.get('matrix:decode_wait'))"
done
done
