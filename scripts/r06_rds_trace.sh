cd $GRAFT_REPO_ROOT && export PYTHONPATH=$PWD
THREADS=16 COPIES=64 KEEP_BAM=/tmp/easy.bam timeout 900 python scripts/bench_bam_ingest.py 250000 > /dev/null 2>&1
DROPEST_RDS_TRACE=1 DROPEST_BAM_DEVICE=1 tests/cpp/bam_to_counts /tmp/res filled 20 100 - 16 /tmp/easy.bam 2>&1 | grep -E "rds|write_ms" | cut -c1-300
ls -la /tmp/res*
