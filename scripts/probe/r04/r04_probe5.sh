#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
T=${1:-r04f}
mkdir -p gpurun_out/$T
B="python bench.py --no-secondary --steps 20 --warmup 3 --cpu-sample 0 --push-sample 0"
export DROPEST_BENCH_NO_FORMS=1
DROPEST_BENCH_MATRIX_FORM=bytes $B > gpurun_out/$T/bench_bytes.json 2>/dev/null
for c in 8 12 24; do DROPEST_WIRE_CHUNKS=$c $B > gpurun_out/$T/bench_chunks_$c.json 2>/dev/null; done
DROPEST_WIRE_SDMA=1 $B > gpurun_out/$T/bench_sdma12.json 2>/dev/null
python -m pytest tests/test_gpu_wire.py -x -q 2>&1 | tail -2
DROPEST_BENCH_MATRIX_FORM=bytes $B > gpurun_out/$T/bench_bytes2.json 2>/dev/null
python - $T <<'P'
import json,glob,sys
for f in sorted(glob.glob('gpurun_out/%s/bench_*.json'%sys.argv[1])):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        h=d['host_stage_wall_ms_per_step']
        print(f.split('/')[-1], d['ms_per_step'], min(d['step_ms']), sorted(d['step_ms'])[len(d['step_ms'])//2], 'ksum', d['roofline']['pipeline']['kernel_ms_per_step'], {k:h[k] for k in h if 'matrix' in k or 'prefetch' in k or k=='set_initialized'})
    except Exception as e: print(f, 'ERR', e)
P
