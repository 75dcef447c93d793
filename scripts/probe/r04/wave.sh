#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
export DROPEST_BENCH_NO_FORMS=1
B="python bench.py --no-secondary --steps 12 --warmup 3 --cpu-sample 0 --push-sample 0"
for m in 0 128 64 128 0; do
if [ $m != 0 ]; then export DROPEST_SS_LOCAL_WAVE=$m; else unset DROPEST_SS_LOCAL_WAVE; fi
$B 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']; s=sorted(d['step_ms'])
print('wave $m', d['ms_per_step'], s[6], 'ksum', d['roofline']['pipeline']['kernel_ms_per_step'], 'ss_local', k.get('ss_local:keys',{}).get('ms_per_step'), 'viol', d.get('stats',{}).get('count:ss_order_violation'))"
done
DROPEST_SS_LOCAL_WAVE=128 timeout 900 python -m pytest tests/test_gpu_ssort.py -x -q 2>&1 | tail -2
