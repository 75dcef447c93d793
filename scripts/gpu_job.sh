#!/bin/bash
# One parametrised job for a gpurun call (replaces the one-off probe scripts of round 4).  Runs from the repo root on the GPU box; everything
# it prints to keep goes to gpurun_out/.  Usage:  bash scripts/gpu_job.sh <word> [args...]   (several jobs: separate them with ---)
#   tests <pytest args>             python -m pytest -m gpu -x -q <args>         -> gpurun_out/tests_<n>.log
#   bench <name> <bench.py args>    python bench.py <args>                       -> gpurun_out/bench_<name>.json (+ a summary line on stdout)
#   stages <name> <bench.py args>   the same, printing the host stages and kernels of the line sorted by time
#   prof <tag>                      scripts/refresh_profiles.sh <tag>
#   py <script> [args]              python <script> [args]                       -> gpurun_out/py_<n>.log
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export PYTHONPATH=$PWD
mkdir -p gpurun_out
n=0
run_one() {
  local word=$1; shift
  n=$((n+1))
  case "$word" in
    tests) timeout 3000 python -m pytest -m gpu -x -q "$@" > gpurun_out/tests_$n.log 2>&1; echo "tests_$n rc $? : $(tail -1 gpurun_out/tests_$n.log)";;
    bench|stages)
      local name=$1; shift
      timeout 1500 python bench.py "$@" 2> gpurun_out/bench_$name.err | tail -1 > gpurun_out/bench_$name.json
      python - "$name" "$word" <<'PY'
import json, sys
name, word = sys.argv[1], sys.argv[2]
try:
    d = json.load(open("gpurun_out/bench_%s.json" % name))
except Exception as e:
    print("bench", name, "no line:", e); sys.exit(0)
print("bench", name, json.dumps(d.get("summary")))
print("   step_ms", d.get("step_ms"))
if word == "stages":
    for title, tab in (("host stages", d.get("host_stage_wall_ms_per_step") or {}), ("kernels", {k: v["ms_per_step"] for k, v in (d.get("kernels_ms_per_step") or {}).items()})):
        print("  ", title, "(sum %.3f)" % sum(tab.values()))
        for k, v in sorted(tab.items(), key=lambda kv: -kv[1])[:40]:
            print("      %-44s %.3f" % (k, v))
PY
      ;;
    prof) bash scripts/refresh_profiles.sh "$@" > gpurun_out/prof_$n.log 2>&1; tail -12 gpurun_out/prof_$n.log;;
    py) timeout 3000 python "$@" > gpurun_out/py_$n.log 2>&1; echo "py_$n rc $?"; tail -30 gpurun_out/py_$n.log;;
    *) echo "unknown job: $word";;
  esac
}
args=()
for a in "$@"; do
  if [ "$a" = "---" ]; then run_one "${args[@]}"; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && run_one "${args[@]}"
