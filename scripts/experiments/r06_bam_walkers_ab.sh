#!/bin/bash
# A/B on one box: the window's block table walked by one thread or by four stretches; files made once, bam_to_counts run alternately
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH=$PWD
D=/tmp/walk_ab; mkdir -p $D
THREADS=16 COPIES=256 KEEP_BAM=$D/easy64.bam timeout 900 python scripts/bench_bam_ingest.py 250000 > /dev/null 2>&1
REAL=1 THREADS=16 COPIES=32 KEEP_BAM=$D/real8.bam timeout 900 python scripts/bench_bam_ingest.py 250000 > /dev/null 2>&1
REAL=1 THREADS=16 COPIES=128 KEEP_BAM=$D/real32.bam timeout 900 python scripts/bench_bam_ingest.py 250000 > /dev/null 2>&1
ls -la $D
for rep in 1 2 3; do
  for f in easy64 real8 real32; do
    for w in 4 1; do
      if [ $w = 1 ]; then export DROPEST_BAM_ONE_WALKER=1; else unset DROPEST_BAM_ONE_WALKER; fi
      r=$(DROPEST_BAM_DEVICE=1 timeout 300 tests/cpp/bam_to_counts $D/res filled 20 100 - 16 $D/$f.bam 2>/dev/null | grep -o '"ingest_ms": [0-9.]*')
      echo "$f walkers=$w $r"
    done
  done
done
rm -rf $D
