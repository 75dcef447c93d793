#!/bin/bash
# Developer check (GPU box, repo root): three BAM files in one run -- the 10.8 x synthetic file, one that deflates 3.2 x (REAL=1), the first again --
# through the host reader and through the device path (the decoder is kept between the files; windows of different sizes; uploads under the
# kernels): the two .rds files must be the same bytes.
export PYTHONPATH=$PWD; O=gpurun_out/two; mkdir -p $O
THREADS=16 COPIES=16 KEEP_BAM=$O/a.bam timeout 900 python scripts/bench_bam_ingest.py 250000 > /dev/null 2>&1
REAL=1 THREADS=16 COPIES=8 KEEP_BAM=$O/b.bam timeout 900 python scripts/bench_bam_ingest.py 250000 > /dev/null 2>&1
for mode in host device; do
  if [ $mode = device ]; then export DROPEST_BAM_DEVICE=1; fi
  tests/cpp/bam_to_counts $O/res_$mode filled 20 100 - 16 $O/a.bam $O/b.bam $O/a.bam 2> $O/err_$mode.txt | tail -1 | cut -c1-200
done
cmp $O/res_host.rds $O/res_device.rds && echo "same rds"
grep -c "device path: [0-9]* windows" $O/err_device.txt
rm -f $O/*.bam $O/res_*
