#!/bin/bash
cd /root/repo
export DROPEST_BENCH_NO_FORMS=1
B="python bench.py --no-secondary --steps 20 --warmup 3 --cpu-sample 0 --push-sample 0"
for i in 1 2 3; do
$B 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_stage_wall_ms_per_step']
c=d['roofline'].get('candidates',{})
print('plain u32', d['ms_per_step'], sorted(d['step_ms'])[10], 'ksum', d['roofline']['pipeline']['kernel_ms_per_step'], 'bk', c.get('build_keys',{}).get('ms_per_step'), 'decode_wait', h.get('matrix:decode_wait'), 'cm', h.get('matrix:cm'))
"
done
DROPEST_BENCH_MATRIX_FORM=bytes $B 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain bytes', d['ms_per_step'], sorted(d['step_ms'])[10])"
$B --sharded 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sharded', d['ms_per_step'], sorted(d['step_ms'])[10])"
DROPEST_WIRE_TRACE=1 $B --steps 3 2>&1 >/dev/null | grep -i -E "numa|node" | head -5
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; lscpu | grep -E "NUMA node|Model name" | head -4
