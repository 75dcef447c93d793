#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
T=${1:-r04r}
mkdir -p gpurun_out/$T
B="python bench.py --no-secondary --steps 120 --warmup 3 --cpu-sample 0 --push-sample 0"
export DROPEST_BENCH_NO_FORMS=1
DROPEST_TAIL_TRACE=1 DROPEST_WIRE_TRACE=1 $B 2> gpurun_out/$T/tail.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=sorted(d['step_ms']); print('wire', d['ms_per_step'], s[60], s[-5:]); print([ (i,x) for i,x in enumerate(d['step_ms']) if x>11])"
python - $T <<'P'
import sys,re
lines=[l.rstrip() for l in open('gpurun_out/%s/tail.err'%sys.argv[1]) if l.startswith('[tail]') or l.startswith('[wire] nnz')]
# group into passes
passes=[];cur=[]
for l in lines:
    if 'flag kernels enqueued' in l and cur: passes.append(cur); cur=[]
    cur.append(l)
passes.append(cur)
def end(p):
    t=[float(re.search(r'\[tail\]\s+([0-9.]+)',l).group(1)) for l in p if l.startswith('[tail]')]
    return max(t) if t else 0
bad=[p for p in passes if end(p)>4500]
print(len(passes), 'passes;', len(bad), 'slow')
for p in bad[:3]: print('\n'.join(p)); print('---')
P
