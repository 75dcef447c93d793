#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_multiproc.py tests/test_gpu_multi.py tests/test_gpu_quality.py tests/test_gpu_bam.py -x -q > gpurun_out/mp_tests.log 2>&1; echo "tests rc $?"; grep -E "passed|failed|error" gpurun_out/mp_tests.log | tail -3
B="python bench.py --no-secondary --steps 20 --warmup 3 --cpu-sample 0 --push-sample 0"
export DROPEST_BENCH_NO_FORMS=1
for i in 1 2; do
$B --sharded 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=sorted(d['step_ms']); h=d['host_stage_wall_ms_per_step']; print('sharded packed', d['ms_per_step'], s[10], {k:v for k,v in h.items() if k.startswith('shard:') and v>0.1})"
DROPEST_SHARD_UNPACK_FIRST=1 $B --sharded 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=sorted(d['step_ms']); print('sharded unpack first', d['ms_per_step'], s[10])"
done
DROPEST_BENCH_MATRIX_FORM=bytes $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=sorted(d['step_ms']); print('plain bytes', d['ms_per_step'], s[10])"
$B --config c4 --reads 1.25e8 --sharded 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 sharded', d['ms_per_step'])"
