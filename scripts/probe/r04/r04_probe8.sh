#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
T=${1:-r04m}
mkdir -p gpurun_out/$T
B="python bench.py --no-secondary --steps 6 --warmup 2 --cpu-sample 0 --push-sample 0"
export DROPEST_BENCH_NO_FORMS=1
DROPEST_TAIL_TRACE=1 $B > gpurun_out/$T/bench.json 2> gpurun_out/$T/tail.err
grep "^\[tail\]" gpurun_out/$T/tail.err | tail -60
DROPEST_TAIL_TRACE=1 DROPEST_BENCH_MATRIX_FORM=bytes $B > gpurun_out/$T/bench_bytes.json 2> gpurun_out/$T/tail_bytes.err
echo BYTES; grep "^\[tail\]" gpurun_out/$T/tail_bytes.err | tail -24
