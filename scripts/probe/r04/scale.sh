#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
DROPEST_MULTI_SCALE=4 timeout 2400 python -m pytest tests/test_gpu_multi.py -x -q -k "not c4_at_its_per_gpu_size and not 2e7" > gpurun_out/scale_multi.log 2>&1; echo "multi x4 rc $?"; tail -3 gpurun_out/scale_multi.log
for off in 1000 2000 3000; do
DROPEST_STRESS_SEED_OFFSET=$off timeout 1800 python -m pytest tests/test_gpu_stress.py -x -q > gpurun_out/scale_stress_$off.log 2>&1; echo "stress +$off rc $?"; tail -2 gpurun_out/scale_stress_$off.log
done
SOAK_FREE_ONLY=1 timeout 1500 python scripts/soak_sharded.py 80 777 4000000 > gpurun_out/soak_free2.log 2>&1; echo "free soak rc $?"; grep -c "^ok" gpurun_out/soak_free2.log; tail -1 gpurun_out/soak_free2.log
