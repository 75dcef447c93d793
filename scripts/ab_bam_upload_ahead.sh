#!/bin/bash
# A/B of the device BAM path with and without the upload of window k + 1 under the kernels of window k (DROPEST_BAM_NO_UPLOAD_AHEAD=1)
# usage (GPU box, repo root): bash scripts/ab_bam_upload_ahead.sh [copies of 1e6 reads, default 64]
R=${GRAFT_REPO_ROOT:-$(pwd)}; export PYTHONPATH=$R; cd $R
OUT=$R/gpurun_out/ab_upload; mkdir -p $OUT
THREADS=4 COPIES=${1:-64} KEEP_BAM=$OUT/synth.bam timeout 1500 python scripts/bench_bam_ingest.py 1000000 2> $OUT/bench.err | cut -c1-420
for rep in 1 2 3; do
  for mode in ahead not_ahead; do
    if [ $mode = not_ahead ]; then export DROPEST_BAM_NO_UPLOAD_AHEAD=1; else unset DROPEST_BAM_NO_UPLOAD_AHEAD; fi
    DROPEST_BAM_DEVICE=1 DROPEST_BAM_TRACE=1 timeout 300 $R/tests/cpp/bam_to_counts $OUT/res filled 20 100 - 16 $OUT/synth.bam 2> $OUT/err.txt | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', 'ingest_ms', d['ingest_ms'])"
    grep "device path" $OUT/err.txt | sed -e "s/.*behind the header; //" -e "s/.*device path: //" | cut -c1-330
  done
done
rm -f $OUT/synth.bam $OUT/res*
