#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r04l}
mkdir -p $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R DROPEST_BENCH_NO_FORMS=1
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/$T/trace -o r -- python $R/bench.py --no-secondary --steps 5 --warmup 2 --cpu-sample 0 --push-sample 0 > $R/gpurun_out/$T/bench_under_rocprof.json 2> $R/gpurun_out/$T/rocprof.err
cd $R
python scripts/trace_gaps.py gpurun_out/$T/trace 40 > gpurun_out/$T/gaps.txt 2>&1
cat gpurun_out/$T/gaps.txt
ls gpurun_out/$T/trace | head
