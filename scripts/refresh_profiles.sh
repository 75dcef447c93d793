#!/bin/bash
# Collects the round's profile artefacts on the GPU box into gpurun_out/prof (copy what is to be judged into profiles/):
#   kernel-trace stats of a 5-step bench run, FETCH_SIZE / WRITE_SIZE counters in separate passes (never together with
#   any other trace domain) over ONE step, the PMC summary (+ profiles/pmc_pipeline.json, which bench.py quotes), the
#   per-kernel roofline table, and bench lines: C2 (default), C2 through the sharded runner, C2 with the LSD sort,
#   C3 at 1e9 reads, the C4 shape plain and through the sharded runner.
# usage (from the repo root on the GPU box): bash scripts/refresh_profiles.sh r02a
set -u
TAG=${1:-r02a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o r -- python $R/bench.py --steps 5 --warmup 2 --cpu-sample 0 --push-sample 0 --no-secondary > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/kt.err
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o r -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --push-sample 0 --no-secondary > /dev/null 2> $OUT/fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o r -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --push-sample 0 --no-secondary > /dev/null 2> $OUT/write.err
cd $R
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
python scripts/pmc_summary.py $OUT/fetch $OUT/write $OUT $TAG 100000000 > $OUT/pmc_summary.log 2>&1
cp $OUT/pmc_pipeline.json profiles/pmc_pipeline.json 2>/dev/null    # so that the bench lines below quote the fresh traffic
python scripts/roofline_table.py $OUT/${TAG}_pmc_summary.csv > $OUT/${TAG}_roofline_by_kernel.md 2> $OUT/roofline.err
if [ -n "${ONLY_PMC:-}" ]; then rm -rf $OUT/kt $OUT/fetch $OUT/write; tail -4 $OUT/pmc_summary.log; exit 0; fi
timeout 600 python bench.py --sharded --cpu-sample 0 2> /dev/null | head -1 > $OUT/${TAG}_bench_c2_sharded_runner.json
DROPEST_SORT=lsd timeout 600 python bench.py --cpu-sample 0 --no-secondary 2> /dev/null | head -1 > $OUT/${TAG}_bench_c2_lsd_sort.json
rm -rf $OUT/kt $OUT/fetch $OUT/write
ls -la $OUT; tail -4 $OUT/pmc_summary.log
timeout 900 python bench.py --config c3 --reads 1e9 --steps 3 --warmup 1 --cpu-sample 3e6 2> /dev/null | head -1 > $OUT/${TAG}_bench_c3_1e9.json
# kernel-trace stats of the C3 pass at 1e9 reads
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt3 -o r -- python $R/bench.py --config c3 --reads 1e9 --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2> $OUT/kt3.err)
cp $(find $OUT/kt3 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_c3_1e9_kernel_stats.csv 2>/dev/null; rm -rf $OUT/kt3
# HBM counters of the C3 pass at 1e9 reads (separate passes, kernel-trace only): profiles/pmc_pipeline_c3_1e9.json, which bench.py quotes for secondary.c3_1e9
(cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch3 -o r -- python $R/bench.py --config c3 --reads 1e9 --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2> $OUT/fetch3.err)
(cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write3 -o r -- python $R/bench.py --config c3 --reads 1e9 --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2> $OUT/write3.err)
DROPEST_PMC_WORKLOAD=c3 DROPEST_PMC_FILE=pmc_pipeline_c3_1e9.json python scripts/pmc_summary.py $OUT/fetch3 $OUT/write3 $OUT ${TAG}_c3_1e9 1000000000 > $OUT/pmc_summary_c3.log 2>&1
cp $OUT/pmc_pipeline_c3_1e9.json profiles/pmc_pipeline_c3_1e9.json 2>/dev/null
rm -rf $OUT/fetch3 $OUT/write3; tail -3 $OUT/pmc_summary_c3.log
timeout 600 python bench.py --config c4 --reads 1.25e8 --steps 5 --warmup 2 --cpu-sample 3e6 2> /dev/null | head -1 > $OUT/${TAG}_bench_c4_1gpu.json
timeout 600 python bench.py --config c4 --reads 1.25e8 --steps 5 --warmup 2 --cpu-sample 0 --sharded 2> /dev/null | head -1 > $OUT/${TAG}_bench_c4_sharded_runner.json
timeout 1200 python bench.py --steps 20 --warmup 5 2> $OUT/bench.err | head -1 > $OUT/${TAG}_bench_c2.json     # the driver's command: C2 + secondary.c3_1e9 + the sharded runner
python - <<PY
import json
for n in ("c2", "c2_sharded_runner", "c2_lsd_sort", "c3_1e9", "c4_1gpu", "c4_sharded_runner"):
    d = json.load(open("$OUT/${TAG}_bench_%s.json" % n)); print(n, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], (d["roofline"].get("pipeline") or {}).get("amplification"))
PY
