#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
export DROPEST_BENCH_NO_FORMS=1
python bench.py --no-secondary --cpu-sample 0 --push-sample 0 --config c3 --reads 1e9 --steps 4 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']
print('c3', d['ms_per_step'], sorted(d['step_ms']), 'ksum', d['roofline']['pipeline']['kernel_ms_per_step'])
for n,v in sorted(k.items(), key=lambda kv:-kv[1]['ms_per_step'])[:34]: print('%8.2f %s'%(v['ms_per_step'],n))"
