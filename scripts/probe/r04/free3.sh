#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_multi.py -x -q -s -k "two_shards_at_2e7" > gpurun_out/free_2e7.log 2>&1; echo "2e7 rc $?"; grep -E "c4_|passed|failed|Error" gpurun_out/free_2e7.log | tail -20
