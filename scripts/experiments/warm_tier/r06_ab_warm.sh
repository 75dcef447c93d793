#!/bin/bash
# round 6: the barcode pass over the L2-resident table of listed barcodes (k_cbhash.h: cb_insert_listed_kernel) against the LDS list of round 5
# (DROPEST_CB_MODE=lds DROPEST_CB_NO_WARM=1), C2 and C3 at BASELINE's sizes; with and without the prefix launch; the fused key pass at C3
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export PYTHONPATH=$PWD
mkdir -p gpurun_out
B="--cpu-sample 0 --push-sample 0 --no-secondary"
C3="--config c3 --reads 1e9 --steps 5 --warmup 1"
export DROPEST_BENCH_NO_FORMS=1 DROPEST_BENCH_NO_BAM=1
show() { python - "$1" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_%s.json" % sys.argv[1]))
k = d["kernels_ms_per_step"]
print("   ", {x: round(k[x]["ms_per_step"], 3) for x in k if any(x.startswith(p) for p in ("cb_insert", "cb_hot", "build_keys", "ss_scatter:L1", "ss_sample")) and k[x]["ms_per_step"] > 0.2})
PY
}
run() { local name=$1; shift; bash scripts/gpu_job.sh bench "$name" "$@" | head -2; show "$name"; }
run c2_l2 $B
DROPEST_CB_PREFIX_DIV=0 run c2_l2_noprefix $B
DROPEST_CB_MODE=lds DROPEST_CB_NO_WARM=1 run c2_lds $B
run c3_l2 $C3 $B
DROPEST_CB_PREFIX_DIV=0 run c3_l2_noprefix $C3 $B
DROPEST_NO_FUSED_KEYS=1 run c3_l2_nofused $C3 $B
DROPEST_CB_MODE=lds DROPEST_CB_NO_WARM=1 run c3_lds $C3 $B
