#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
export DROPEST_BENCH_NO_FORMS=1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "ordering or device_ordering or max_cells or whitelist" > gpurun_out/c3b_tests.log 2>&1; echo "rc $?"; tail -1 gpurun_out/c3b_tests.log
B="python bench.py --no-secondary --cpu-sample 0 --push-sample 0"
for i in 1 2; do
$B --config c3 --reads 1e9 --steps 4 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_stage_wall_ms_per_step']
print('c3', d['ms_per_step'], sorted(d['step_ms']), {k:v for k,v in h.items() if k.startswith('sort_filtered')})"
done
