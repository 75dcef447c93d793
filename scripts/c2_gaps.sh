#!/bin/bash
# C2: where the device idles inside one timed pass (kernel trace -> scripts/trace_gaps.py).  Output: gpurun_out/c2_gaps.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/c2ev
mkdir -p $OUT
export PYTHONPATH=$R TMPDIR=/tmp
(cd $R && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o r -- python $R/bench.py --steps 4 --warmup 2 --cpu-sample 0 --push-sample 0 --no-secondary > $OUT/bench_under_trace.json 2> $OUT/kt.err)
TRACE_PASS=${TRACE_PASS:-4} python $R/scripts/trace_gaps.py $OUT/kt 70 > $R/gpurun_out/c2_gaps.txt 2>&1
rm -rf $OUT/kt
head -45 $R/gpurun_out/c2_gaps.txt
