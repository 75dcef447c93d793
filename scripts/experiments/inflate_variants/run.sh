#!/bin/bash
# kernel experiments of csrc/k_inflate.h: variant builds of bgzf_api.hip (waves per SIMD, the fence before a match copy) through scripts/bench_bgzf_inflate.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH=$PWD
for f in scripts/experiments/inflate_variants/libbgzf_*.so; do
  echo "== $f"; DROPEST_BGZF_LIB=$PWD/$f timeout 300 python scripts/bench_bgzf_inflate.py 300000 40 2>&1 | tail -1
done
