#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
T=${1:-r04v}
mkdir -p gpurun_out/$T
B="python bench.py --no-secondary --steps 10 --warmup 3 --cpu-sample 0 --push-sample 0"
export DROPEST_BENCH_NO_FORMS=1
$B --sharded > gpurun_out/$T/bench_sharded.json 2> gpurun_out/$T/sharded.err
$B > gpurun_out/$T/bench_plain.json 2>/dev/null
python - $T <<'P'
import json,sys
a=json.loads(open('gpurun_out/%s/bench_plain.json'%sys.argv[1]).read().strip().splitlines()[-1])
b=json.loads(open('gpurun_out/%s/bench_sharded.json'%sys.argv[1]).read().strip().splitlines()[-1])
print('plain', a['ms_per_step'], 'sharded', b['ms_per_step'])
ka,kb=a['kernels_ms_per_step'],b['kernels_ms_per_step']
for k in sorted(set(ka)|set(kb), key=lambda x:-(kb.get(x,{}).get('ms_per_step',0))):
    x,y=ka.get(k,{}).get('ms_per_step',0),kb.get(k,{}).get('ms_per_step',0)
    if abs(x-y)>0.02: print('  %-32s plain %.3f sharded %.3f'%(k,x,y))
print(sum(v['ms_per_step'] for v in ka.values()), sum(v['ms_per_step'] for v in kb.values()))
hb=b['host_stage_wall_ms_per_step']
print({k:v for k,v in hb.items() if v>0.1})
P
