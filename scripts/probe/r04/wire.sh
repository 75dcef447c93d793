#!/bin/bash
cd /root/repo
export DROPEST_BENCH_NO_FORMS=1
B="python bench.py --no-secondary --steps 12 --warmup 3 --cpu-sample 0 --push-sample 0"
DROPEST_WIRE_TRACE=1 $B 2> gpurun_out/wire.err | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_stage_wall_ms_per_step']; print('plain u32', d['ms_per_step'], sorted(d['step_ms']), h.get('matrix:decode_wait'), h.get('matrix:cm'))"
grep "\[wire\] nnz" gpurun_out/wire.err | tail -12
grep "\[wire\] nodes" gpurun_out/wire.err | tail -2
DROPEST_BENCH_MATRIX_FORM=bytes $B 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain bytes', d['ms_per_step'], sorted(d['step_ms'])[6])"
for t in 8 14 20; do
DROPEST_DECODE_THREADS=$t $B 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_stage_wall_ms_per_step']; print('threads $t', d['ms_per_step'], sorted(d['step_ms'])[6], h.get('matrix:decode_wait'))"
done
DROPEST_DECODE_NUMA=0 $B 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_stage_wall_ms_per_step']; print('numa off', d['ms_per_step'], sorted(d['step_ms'])[6], h.get('matrix:decode_wait'))"
uptime; cat /sys/fs/cgroup/cpu.max; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null | head -2; cat /proc/pressure/cpu 2>/dev/null | head -2
