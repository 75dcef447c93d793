#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
T=${1:-r04d}
mkdir -p gpurun_out/$T
B="python bench.py --no-secondary --steps 10 --warmup 3 --cpu-sample 0 --push-sample 0"
export DROPEST_BENCH_NO_FORMS=1
$B > gpurun_out/$T/bench_a512nt_12.json 2>/dev/null
DROPEST_DECODE_THREADS=15 $B > gpurun_out/$T/bench_a512nt_15.json 2>/dev/null
DROPEST_DECODE_THREADS=8 $B > gpurun_out/$T/bench_a512nt_8.json 2>/dev/null
DROPEST_DECODE_NT=0 $B > gpurun_out/$T/bench_a512plain_12.json 2>/dev/null
DROPEST_DECODE=avx2 $B > gpurun_out/$T/bench_avx2plain_12.json 2>/dev/null
DROPEST_DECODE=avx2 DROPEST_DECODE_THREADS=15 $B > gpurun_out/$T/bench_avx2plain_15.json 2>/dev/null
DROPEST_BENCH_MATRIX_FORM=bytes $B > gpurun_out/$T/bench_bytes.json 2>/dev/null
python - $T <<'P'
import json,glob,sys
for f in sorted(glob.glob('gpurun_out/%s/bench_*.json'%sys.argv[1])):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        h=d['host_stage_wall_ms_per_step']
        print(f.split('/')[-1], d['ms_per_step'], min(d['step_ms']), 'ksum', d['roofline']['pipeline']['kernel_ms_per_step'], {k:h[k] for k in h if 'matrix' in k or 'prefetch' in k or k=='set_initialized'})
    except Exception as e: print(f, 'ERR', e)
P
