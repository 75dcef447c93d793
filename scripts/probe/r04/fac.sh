#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
timeout 1500 python -m pytest tests/test_gpu_facade.py tests/test_gpu_bam.py -x -q > gpurun_out/fac.log 2>&1; echo "rc $?"; tail -12 gpurun_out/fac.log
