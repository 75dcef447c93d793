#!/bin/bash
# rocprofv3 kernel-trace stats of the device BAM path (csrc/k_inflate.h, k_bamparse.h) through tests/cpp/bam_to_counts on a synthetic 10x BAM,
# and the inflate kernel alone through scripts/bench_bgzf_inflate.py.  Output: gpurun_out/prof_bam/<tag>_bam_kernel_stats.csv + the bench lines.
# usage (GPU box, repo root): bash scripts/prof_bam_device.sh r05d
set -u
TAG=${1:-r05d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_bam
mkdir -p $OUT
export PYTHONPATH=$R
cd $R
timeout 600 python scripts/bench_bgzf_inflate.py 300000 40 2> /dev/null | tail -1 > $OUT/${TAG}_bgzf_inflate.json
THREADS=16 COPIES=16 KEEP_BAM=$OUT/synth.bam timeout 1200 python scripts/bench_bam_ingest.py 1000000 > $OUT/${TAG}_bam_ingest.jsonl 2> $OUT/${TAG}_bam_ingest.err
grep "device path" $OUT/${TAG}_bam_ingest.err > $OUT/${TAG}_bam_device_trace.txt
if [ -f $OUT/synth.bam ]; then
  (cd /tmp && export TMPDIR=/tmp && DROPEST_BAM_DEVICE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o r -- $R/tests/cpp/bam_to_counts $OUT/res filled 20 100 - 16 $OUT/synth.bam > $OUT/kt.out 2> $OUT/kt.err)
  cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bam_kernel_stats.csv 2>/dev/null
  rm -rf $OUT/kt $OUT/synth.bam $OUT/res*
fi
cat $OUT/${TAG}_bgzf_inflate.json; cat $OUT/${TAG}_bam_ingest.jsonl | cut -c1-400; head -12 $OUT/${TAG}_bam_kernel_stats.csv
