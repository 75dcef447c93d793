#!/bin/bash
# one environment variable's values in turn on one box: files made once, bam_to_counts run alternately.  usage: r06_bam_env_ab.sh VAR "v1 v2 ..." [reps]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH=$PWD
VAR=$1; VALS=$2; REPS=${3:-3}
D=/tmp/env_ab; mkdir -p $D
THREADS=16 COPIES=64 KEEP_BAM=$D/easy16.bam timeout 900 python scripts/bench_bam_ingest.py 250000 > /dev/null 2>&1
REAL=1 THREADS=16 COPIES=32 KEEP_BAM=$D/real8.bam timeout 900 python scripts/bench_bam_ingest.py 250000 > /dev/null 2>&1
for rep in $(seq $REPS); do
  for f in real8 easy16; do
    for v in $VALS; do
      if [ "$v" = default ]; then unset $VAR; else export $VAR="$v"; fi
      r=$(DROPEST_BAM_DEVICE=1 timeout 300 tests/cpp/bam_to_counts $D/res filled 20 100 - 16 $D/$f.bam 2>/dev/null | grep -o '"ingest_ms": [0-9.]*')
      echo "$f $VAR=$v $r"
    done
  done
done
rm -rf $D
