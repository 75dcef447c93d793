#!/bin/bash
# C3 at 1e9 reads: where the device idles inside one pass (kernel trace -> scripts/trace_gaps.py) and the SQ / TCC counters of the four
# largest kernels (scripts/pmc_kernels.sh over bench.py at C3 size).  Output: gpurun_out/c3ev/{gaps.txt,sq_counters.txt}
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/c3ev
mkdir -p $OUT
export PYTHONPATH=$R TMPDIR=/tmp
CMD="python $R/bench.py --config c3 --reads 1e9 --steps 2 --warmup 1 --cpu-sample 0 --no-secondary"
(cd $R && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o r -- $CMD > $OUT/bench_under_trace.json 2> $OUT/kt.err)
TRACE_PASS=2 python $R/scripts/trace_gaps.py $OUT/kt 40 > $OUT/gaps.txt 2>&1
rm -rf $OUT/kt
PMC_CMD="python $R/bench.py --config c3 --reads 1e9 --steps 1 --warmup 0 --cpu-sample 0 --no-secondary" bash $R/scripts/pmc_kernels.sh "cb_insert|build_keys|ss_scatter_res|ss_local_kernel|ss_compact_cg" > $OUT/sq_counters.txt 2>&1
rm -rf $R/gpurun_out/pmc
head -30 $OUT/gaps.txt; head -60 $OUT/sq_counters.txt
