#!/bin/bash
# round 6: cb_insert_listed_kernel -- rounds of the listed table before a read falls through to the big table x room of the listed table
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export PYTHONPATH=$PWD
mkdir -p gpurun_out
export DROPEST_BENCH_NO_FORMS=1 DROPEST_BENCH_NO_BAM=1
one() { # name, bench args...
  local name=$1; shift
  python bench.py --cpu-sample 0 --push-sample 0 --no-secondary --steps 3 --warmup 1 "$@" 2> gpurun_out/sw_$name.err | tail -1 > gpurun_out/sw_$name.json
  python - "$name" <<'PY'
import json, sys
try:
    d = json.load(open("gpurun_out/sw_%s.json" % sys.argv[1]))
    k = d["kernels_ms_per_step"]
    print("%-28s step %.2f  kernels %.2f  " % (sys.argv[1], d["ms_per_step"], d["roofline"]["pipeline"]["kernel_ms_per_step"]),
          {x: round(k[x]["ms_per_step"], 3) for x in k if any(x.startswith(p) for p in ("cb_insert", "build_keys", "ss_scatter:L1"))})
except Exception as e:
    print(sys.argv[1], "failed:", e)
PY
}
for cfg in c2 c3; do
  if [ $cfg = c3 ]; then A="--config c3 --reads 1e9"; else A=""; fi
  DROPEST_CB_MODE=lds DROPEST_CB_NO_WARM=1 one ${cfg}_lds $A
  for room in 2 4 8; do for rounds in 1 2 3; do
    DROPEST_CB_WARM_ROOM=$room DROPEST_CB_LISTED_ROUNDS=$rounds one ${cfg}_room${room}_rounds${rounds} $A
  done; done
done
