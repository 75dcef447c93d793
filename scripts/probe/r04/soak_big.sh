#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out/soak
timeout 2400 python scripts/soak_sharded.py 50 31337 25000000 > gpurun_out/soak/r04_soak_sharded_50_streams_to_2.5e7.log 2>&1; echo "sharded big rc $?"; grep -c "^ok" gpurun_out/soak/r04_soak_sharded_50_streams_to_2.5e7.log; tail -1 gpurun_out/soak/r04_soak_sharded_50_streams_to_2.5e7.log
SEED=77 ITERS=10 NMAX=250000000 timeout 1500 python scripts/soak_paths.py > gpurun_out/soak/r04_soak_paths_10_streams_to_2.5e8.log 2>&1; echo "paths rc $?"; tail -1 gpurun_out/soak/r04_soak_paths_10_streams_to_2.5e8.log | cut -c1-200
