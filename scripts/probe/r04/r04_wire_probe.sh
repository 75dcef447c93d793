#!/bin/bash
# round 4: the 32-bit slots over the byte wire -- tests first, then the C2 step under a few decoder settings
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04a
python -m pytest tests/test_gpu_wire.py tests/test_gpu_narrow.py -x -q -m gpu > gpurun_out/r04a/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r04a/tests.log
tail -3 gpurun_out/r04a/tests.log
nproc; cat /sys/fs/cgroup/cpu.max
B="python bench.py --no-secondary --steps 10 --warmup 3 --cpu-sample 0 --push-sample 0"
$B > gpurun_out/r04a/bench_default.json 2> gpurun_out/r04a/bench_default.err
for t in 4 8 16 24; do DROPEST_DECODE_THREADS=$t DROPEST_BENCH_NO_FORMS=1 $B > gpurun_out/r04a/bench_thr$t.json 2>/dev/null; done
DROPEST_WIRE_CHUNKS=6 DROPEST_BENCH_NO_FORMS=1 $B > gpurun_out/r04a/bench_chunks6.json 2>/dev/null
DROPEST_WIRE_CHUNKS=24 DROPEST_BENCH_NO_FORMS=1 $B > gpurun_out/r04a/bench_chunks24.json 2>/dev/null
DROPEST_DECODE_NT=1 DROPEST_BENCH_NO_FORMS=1 $B > gpurun_out/r04a/bench_nt.json 2>/dev/null
DROPEST_DECODE_SCALAR=1 DROPEST_BENCH_NO_FORMS=1 $B > gpurun_out/r04a/bench_scalar.json 2>/dev/null
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04a/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['ms_per_step'], d['step_ms'], d['config'].get('matrix_forms'), {k:v for k,v in d['host_stage_wall_ms_per_step'].items() if 'matrix' in k or 'decode' in k or 'prefetch' in k})
    except Exception as e: print(f, 'ERR', e)
P
