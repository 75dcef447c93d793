#!/bin/bash
cd /root/repo
export DROPEST_BENCH_NO_FORMS=1
B="python bench.py --no-secondary --steps 20 --warmup 3 --cpu-sample 0 --push-sample 0"
$B --sharded 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_stage_wall_ms_per_step']
print('sharded', d['ms_per_step'], sorted(d['step_ms'])[10])
for k,v in sorted(h.items(), key=lambda kv:-kv[1])[:40]: print('%8.3f  %s'%(v,k))
ks=d.get('kernel_ms_per_step') or {}
for k,v in sorted(ks.items(), key=lambda kv:-kv[1])[:30]: print('   k %8.3f  %s'%(v,k))
"
DROPEST_BENCH_MATRIX_FORM=bytes $B 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_stage_wall_ms_per_step']
print('plain bytes', d['ms_per_step'], sorted(d['step_ms'])[10])
for k,v in sorted(h.items(), key=lambda kv:-kv[1])[:25]: print('%8.3f  %s'%(v,k))
"
