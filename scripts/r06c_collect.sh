#!/bin/bash
# round 6c: what changed after the r06b set -- the lane-parallel inflate and the BAM path -- measured again: the driver's bench line (its
# secondary.bam_ingest carries both files), BAM -> container traces and kernel stats, the inflate kernel alone at 50 360 blocks on both files,
# its phases (-DINFP_PROFILE build) and SQ / TCC counters, and a soak of random BAM files host reader against device
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export PYTHONPATH=$PWD
OUT=gpurun_out/${TAG:-r06c}; T=${TAG:-r06c}; mkdir -p $OUT
timeout 1200 python bench.py --steps 20 --warmup 5 2> $OUT/bench.err | head -1 > $OUT/${T}_bench_c2.json
bash scripts/r06_bam_breakdown.sh > $OUT/breakdown.log 2>&1
for k in real:3x easy:10x; do a=${k%%:*}; b=${k#*:}; cp gpurun_out/bam_r06/${a}_device_trace.txt $OUT/${T}_bam_${b}_file_device_trace.txt; cp gpurun_out/bam_r06/${a}_ingest.jsonl $OUT/${T}_bam_${b}_file_ingest.jsonl; cp gpurun_out/bam_r06/${a}_kernel_stats.csv $OUT/${T}_bam_${b}_file_kernel_stats.csv 2>/dev/null; done
{ REAL=1 timeout 300 python scripts/bench_bgzf_inflate.py 300000 40 2>&1 | tail -1; timeout 300 python scripts/bench_bgzf_inflate.py 300000 40 2>&1 | tail -1; REAL=1 DROPEST_INFLATE_PAR=0 timeout 300 python scripts/bench_bgzf_inflate.py 300000 40 2>&1 | tail -1; DROPEST_INFLATE_PAR=0 timeout 300 python scripts/bench_bgzf_inflate.py 300000 40 2>&1 | tail -1; } > $OUT/${T}_bgzf_inflate.jsonl
if [ -f scripts/experiments/inflate_variants/libbgzf_par_prof.so ]; then DROPEST_BGZF_LIB=$PWD/scripts/experiments/inflate_variants/libbgzf_par_prof.so timeout 400 python scripts/experiments/inflate_par_profile.py 100000 > $OUT/${T}_inflate_par_phases.txt 2>&1; fi
REAL=1 PMC_CMD="python scripts/bench_bgzf_inflate.py 300000 12" timeout 1200 bash scripts/pmc_kernels.sh "bgzf_inflate_par" > $OUT/${T}_inflate_par_pmc_3x_file.txt 2>&1
timeout 1500 python scripts/soak_bam_device.py 16 > $OUT/${T}_soak_bam_device_16_files.log 2>&1
tail -3 $OUT/${T}_soak_bam_device_16_files.log; cut -c1-400 $OUT/${T}_bench_c2.json; cat $OUT/${T}_bgzf_inflate.jsonl | cut -c1-420
