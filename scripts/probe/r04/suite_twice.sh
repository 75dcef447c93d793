#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/suite
for i in 1 2; do
python -m pytest tests -x -q -m gpu > gpurun_out/suite/run$i.log 2>&1; echo "run $i rc $?"; grep -E "passed|failed|error|core|Fatal" gpurun_out/suite/run$i.log | tail -3
done
python -c "import __graft_entry__ as g; g.smoke()"; echo "smoke rc $?"
