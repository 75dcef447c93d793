#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
T=${1:-r04c}
mkdir -p gpurun_out/$T
python -m pytest tests/test_gpu_ssort.py tests/test_gpu_stress.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/$T/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/$T/tests.log
tail -5 gpurun_out/$T/tests.log
B="python bench.py --no-secondary --steps 10 --warmup 3 --cpu-sample 0 --push-sample 0"
DROPEST_BENCH_NO_FORMS=1 $B > gpurun_out/$T/bench_reserve.json 2> gpurun_out/$T/bench_reserve.err
DROPEST_SS_NO_RESERVE=1 DROPEST_BENCH_NO_FORMS=1 $B > gpurun_out/$T/bench_count.json 2>/dev/null
DROPEST_BENCH_NO_FORMS=1 $B --config c3 --reads 1e9 --steps 3 --warmup 1 > gpurun_out/$T/bench_c3_reserve.json 2>/dev/null
DROPEST_SS_NO_RESERVE=1 DROPEST_BENCH_NO_FORMS=1 $B --config c3 --reads 1e9 --steps 3 --warmup 1 > gpurun_out/$T/bench_c3_count.json 2>/dev/null
python - $T <<'P'
import json,glob,sys
for f in sorted(glob.glob('gpurun_out/%s/bench_*.json'%sys.argv[1])):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d['kernels_ms_per_step']
        print(f.split('/')[-1], d['ms_per_step'], 'ksum', d['roofline']['pipeline']['kernel_ms_per_step'], {x:k[x] for x in k if x.startswith('ss_') and k[x]>0.03})
    except Exception as e: print(f, 'ERR', e)
P
