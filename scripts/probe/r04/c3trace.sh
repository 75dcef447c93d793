#!/bin/bash
R=/root/repo
T=c3gaps
mkdir -p $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R DROPEST_BENCH_NO_FORMS=1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$T/trace -o r -- python $R/bench.py --no-secondary --config c3 --reads 1e9 --steps 2 --warmup 1 --cpu-sample 0 --push-sample 0 > $R/gpurun_out/$T/bench.json 2> $R/gpurun_out/$T/rocprof.err
cd $R
python scripts/trace_gaps.py gpurun_out/$T/trace 10 > gpurun_out/$T/gaps.txt 2>&1
head -40 gpurun_out/$T/gaps.txt
