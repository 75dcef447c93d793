#!/bin/bash
# SQ / TCC counters of selected kernels over one set_initialized at C2 size (separate --pmc passes, kernel-trace only).
# usage on the GPU box: bash scripts/pmc_kernels.sh "cb_insert|build_keys|ss_local"
# PMC_CMD="python bench.py --config c3 --reads 1e9 --steps 1 --warmup 0 --cpu-sample 0 --no-secondary" takes the counters over another command (C3 size)
set -u
PAT=${1:-cb_insert}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_EA0_RDREQ_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rm -rf $OUT/p$i
  (cd $R && VARIANTS=0 timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -o r -- ${PMC_CMD:-python $R/scripts/ss_probe.py} > $OUT/p$i.log 2>&1)
done
cd $R
python - "$PAT" <<'PY'
import csv, glob, re, sys, collections
pat = re.compile(sys.argv[1])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if pat.search(k):
            agg[re.sub(r"\(.*", "", k)[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print(k)
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if all(c in m for c in ("SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "TCC_HIT_sum", "TCC_MISS_sum", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE")):
        wc = m["SQ_WAVE_CYCLES"]
        print("   wait_any %.0f%%  wait_inst %.0f%%  active %.0f%%  lds_conflict %.2f%%  waves %d  tcc_hit %.0f%%  EA reads %.3g (x 64 B = %.1f GB)  EA writes %.3g" % (
            100 * m["SQ_WAIT_ANY"] / wc, 100 * m["SQ_WAIT_INST_ANY"] / wc, 100 * m["SQ_ACTIVE_INST_ANY"] / wc, 100 * m["SQ_LDS_BANK_CONFLICT"] / (256 * m["GRBM_GUI_ACTIVE"]),
            m.get("SQ_WAVES", 0), 100 * m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), m.get("TCC_EA0_RDREQ_sum", 0), m.get("TCC_EA0_RDREQ_sum", 0) * 64 / 1e9, m.get("TCC_EA0_WRREQ_sum", 0)))
    for c, v in sorted(d.items()):
        print("   %-24s %.4g (x%d launches, mean)" % (c, sum(v) / len(v), len(v)))
PY
