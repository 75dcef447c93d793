#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export DROPEST_BENCH_NO_FORMS=1
python bench.py --no-secondary --config c3 --reads 1e9 --steps 5 --warmup 1 --cpu-sample 0 --push-sample 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', d['ms_per_step'], d['step_ms']); h=d['host_stage_wall_ms_per_step']
for k,v in sorted(h.items(), key=lambda kv:-kv[1])[:16]: print('  %-40s %.2f'%(k,v))"
