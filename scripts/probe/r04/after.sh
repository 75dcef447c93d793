#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
export DROPEST_BENCH_NO_FORMS=1
B="python bench.py --no-secondary --steps 12 --warmup 3 --cpu-sample 0 --push-sample 0"
mem() { for n in 0 1; do grep -E "MemFree|FilePages|Mlocked|Unevictable" /sys/devices/system/node/node$n/meminfo | tr '\n' ' '; echo; done; }
run() { DROPEST_WIRE_TRACE=1 $B 2> gpurun_out/after.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_stage_wall_ms_per_step']; print('$1 u32', d['ms_per_step'], sorted(d['step_ms'])[6], 'decode_wait', h.get('matrix:decode_wait'))"
grep "\[wire\] nodes" gpurun_out/after.err | tail -1; grep "\[wire\] nnz 1788" gpurun_out/after.err | tail -2; }
echo "== fresh"; mem; run fresh
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "full_size or 1e8 or 1e9" > gpurun_out/after_tests.log 2>&1; tail -1 gpurun_out/after_tests.log
echo "== after big tests"; mem; run after_big
timeout 1200 python -m pytest tests/test_gpu_multi.py tests/test_gpu_bam.py tests/test_gpu_facade.py -q > gpurun_out/after_tests2.log 2>&1; tail -1 gpurun_out/after_tests2.log
echo "== after multi/bam/facade"; mem; run after_multi
ls /dev/shm | head; df -h /dev/shm | tail -1
