#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_multi.py tests/test_gpu_multiproc.py tests/test_gpu_stress.py -x -q -k "four_parts or whitelist" > gpurun_out/ovf.log 2>&1; echo "rc $?"; tail -30 gpurun_out/ovf.log
