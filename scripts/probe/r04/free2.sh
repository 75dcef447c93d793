#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_multi.py -x -q -k "qualities_follow" > gpurun_out/free_qual.log 2>&1; echo "qual rc $?"; tail -30 gpurun_out/free_qual.log
timeout 1500 python -m pytest tests/test_gpu_multiproc.py -x -q > gpurun_out/free_mp.log 2>&1; echo "mp rc $?"; tail -30 gpurun_out/free_mp.log
