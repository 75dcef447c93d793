#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/soak
export PYTHONPATH=.
SEED=47 ITERS=6 NMAX=160000000 timeout 900 python scripts/soak_paths.py > gpurun_out/soak/r04_soak_paths_8_streams_to_1.6e8.log 2>&1; echo "paths rc $?"; tail -2 gpurun_out/soak/r04_soak_paths_8_streams_to_1.6e8.log
timeout 900 python scripts/soak_sharded.py 50 20261002 3000000 > gpurun_out/soak/r04_soak_sharded_60_streams_all_merges.log 2>&1; echo "sharded rc $?"; tail -2 gpurun_out/soak/r04_soak_sharded_40_streams.log
SOAK_CASES=16 SOAK_SEED=103 timeout 900 python scripts/soak_merge_mid.py > gpurun_out/soak/r04_soak_merge_mid.log 2>&1; echo "merge rc $?"; tail -2 gpurun_out/soak/r04_soak_merge_mid.log
