#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
timeout 2400 python -m pytest tests/test_gpu_ssort.py tests/test_gpu_parity.py -x -q > gpurun_out/os4_tests.log 2>&1; echo "rc $?"; tail -2 gpurun_out/os4_tests.log
