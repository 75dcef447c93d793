#!/bin/bash
# blocks per window (DROPEST_BAM_WINDOW_BLOCKS) on one box: files made once, bam_to_counts run in turn
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH=$PWD
D=/tmp/wb; mkdir -p $D
THREADS=16 COPIES=64 KEEP_BAM=$D/easy16.bam timeout 900 python scripts/bench_bam_ingest.py 250000 > /dev/null 2>&1
REAL=1 THREADS=16 COPIES=32 KEEP_BAM=$D/real8.bam timeout 900 python scripts/bench_bam_ingest.py 250000 > /dev/null 2>&1
REAL=1 THREADS=16 COPIES=128 KEEP_BAM=$D/real32.bam timeout 900 python scripts/bench_bam_ingest.py 250000 > /dev/null 2>&1
for rep in 1 2; do
  for f in real8 easy16 real32; do
    for n in default 2048 4096 6144 8192 12288 16384; do
      if [ $n = default ]; then unset DROPEST_BAM_WINDOW_BLOCKS; else export DROPEST_BAM_WINDOW_BLOCKS=$n; fi
      r=$(DROPEST_BAM_DEVICE=1 timeout 300 tests/cpp/bam_to_counts $D/res filled 20 100 - 16 $D/$f.bam 2>/dev/null | grep -o '"ingest_ms": [0-9.]*')
      echo "$f blocks=$n $r"
    done
  done
done
rm -rf $D
