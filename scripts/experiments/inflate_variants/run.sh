#!/bin/bash
# kernel experiments of csrc/k_inflate.h: variant builds of bgzf_api.hip + annotation_api.hip (hipcc -shared, -D switches) through
# scripts/bench_bgzf_inflate.py, on the 10.8 x synthetic BAM and (REAL=1) on one with random bases and binned qualities
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH=$PWD
for f in scripts/experiments/inflate_variants/libbgzf_*.so; do
  echo "== $f"; DROPEST_BGZF_LIB=$PWD/$f timeout 300 python scripts/bench_bgzf_inflate.py 300000 40 2>&1 | tail -1
  REAL=1 DROPEST_BGZF_LIB=$PWD/$f timeout 300 python scripts/bench_bgzf_inflate.py 300000 12 2>&1 | tail -1
done
