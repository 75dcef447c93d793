#!/bin/bash
# round 6: how a whitelist merge at C3 size spreads over the molecule table (DROPEST_MP_TRACE): the rows that change their key, the tiles they
# sit in, and the share of the table the RECEIVING cells hold -- what "re-aggregate only what a merge changed" could leave untouched
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export PYTHONPATH=$PWD
DROPEST_MP_TRACE=1 DROPEST_BENCH_NO_FORMS=1 python bench.py --config c3 --reads 1e9 --steps 1 --warmup 0 --cpu-sample 0 --push-sample 0 --no-secondary 2>&1 >/dev/null | grep "^\[mp\]" | head -4
