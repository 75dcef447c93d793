#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
T=${1:-r04i}
mkdir -p gpurun_out/$T
B="python bench.py --no-secondary --steps 5 --warmup 2 --cpu-sample 0 --push-sample 0"
export DROPEST_BENCH_NO_FORMS=1
$B > gpurun_out/$T/bench_ref.json 2>/dev/null
DROPEST_SS_PROBE=2 $B > gpurun_out/$T/bench_nopf.json 2>/dev/null
python -m pytest tests/test_gpu_ssort.py -x -q 2>&1 | tail -2
python - $T <<'P'
import json,glob,sys
for f in sorted(glob.glob('gpurun_out/%s/bench_*.json'%sys.argv[1])):
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']
    print(f.split('/')[-1], d['ms_per_step'], {x:k[x]['ms_per_step'] for x in k if x.startswith('ss_scatter')})
P
