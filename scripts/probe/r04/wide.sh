#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_wide_keys.py tests/test_gpu_quality.py -x -q > gpurun_out/wide.log 2>&1; echo "rc $?"; tail -25 gpurun_out/wide.log
