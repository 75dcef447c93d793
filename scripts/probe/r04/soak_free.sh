#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
SOAK_FREE_ONLY=1 timeout 900 python scripts/soak_sharded.py 40 4101 2000000 > gpurun_out/soak_free.log 2>&1; echo "free rc $?"; grep -c "^ok" gpurun_out/soak_free.log; grep -E "^FAIL|^skip|failures|Error|error" gpurun_out/soak_free.log | head -20
timeout 900 python scripts/soak_sharded.py 30 4102 2000000 > gpurun_out/soak_all.log 2>&1; echo "all rc $?"; grep -c "^ok" gpurun_out/soak_all.log; grep -E "^FAIL|^skip|failures|Error|error" gpurun_out/soak_all.log | head -20
