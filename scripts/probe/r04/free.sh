#!/bin/bash
# sharded whitelist-free merges: the new tests, then the files that share code with them
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_multi.py -x -q -k "simple_merge or merge_all or free_merges or poisson or whitelist" > gpurun_out/free_new.log 2>&1; echo "new rc $?"; tail -30 gpurun_out/free_new.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "simple or merge_all or poisson" > gpurun_out/free_parity.log 2>&1; echo "parity rc $?"; tail -5 gpurun_out/free_parity.log
