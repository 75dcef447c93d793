#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ssort.py tests/test_gpu_reuse.py tests/test_gpu_narrow.py tests/test_golden.py -x -q > gpurun_out/quick.log 2>&1; echo "rc $?"; tail -3 gpurun_out/quick.log
