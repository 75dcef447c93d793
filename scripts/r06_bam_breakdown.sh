#!/bin/bash
# round 6: where the time of BAM -> container -> .rds goes on the file that deflates 3.2 x (REAL=1) and on the 10.8 x one; device trace, kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export PYTHONPATH=$PWD
OUT=gpurun_out/bam_r06; mkdir -p $OUT
for kind in real easy; do
  if [ $kind = real ]; then export REAL=1; C=32; else unset REAL; C=64; fi
  THREADS=16 COPIES=$C KEEP_BAM=$OUT/$kind.bam timeout 900 python scripts/bench_bam_ingest.py 250000 > $OUT/${kind}_ingest.jsonl 2> $OUT/${kind}_ingest.err
  grep -E "device path|\[bam\]" $OUT/${kind}_ingest.err > $OUT/${kind}_device_trace.txt
  cat $OUT/${kind}_ingest.jsonl | cut -c1-600
  cat $OUT/${kind}_device_trace.txt | cut -c1-900
  if [ -f $OUT/$kind.bam ]; then
    (cd /tmp && export TMPDIR=/tmp && DROPEST_BAM_DEVICE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/kt_$kind -o r -- $GRAFT_REPO_ROOT/tests/cpp/bam_to_counts /tmp/res filled 20 100 - 16 $GRAFT_REPO_ROOT/$OUT/$kind.bam > /dev/null 2>&1)
    f=$(find $OUT/kt_$kind -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/${kind}_kernel_stats.csv && head -8 $OUT/${kind}_kernel_stats.csv
    rm -rf $OUT/$kind.bam $OUT/kt_$kind
  fi
done
