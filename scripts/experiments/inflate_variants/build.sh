#!/bin/bash
# Builds the variants of the inflate kernel that the tree still carries as -D switches (csrc/k_inflate.h: INF_WAVES_PER_EU, INF_NO_FENCE) into
# scripts/experiments/inflate_variants/libbgzf_*.so, for run.sh.  The other variants the round-5 notes name (funnel-shift bit reader, scalar state,
# arithmetic base tables, LDS history, uniform wave number) were source edits that lost and were removed; their measurements are in
# profiles/history/r05[f-q]_inflate_*.txt.
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=scripts/experiments/inflate_variants
for v in "par_prof:-DINFP_PROFILE" "par_c256:-DINFP_CHUNK=256" "par_c1024:-DINFP_CHUNK=1024" "par_w3:-DINFP_WAVES_PER_EU=3" "w4:-DINF_WAVES_PER_EU=4" "w5:-DINF_WAVES_PER_EU=5" "w6:-DINF_WAVES_PER_EU=6" "w8:-DINF_WAVES_PER_EU=8" "w8_nofence:-DINF_WAVES_PER_EU=8 -DINF_NO_FENCE"; do
  name=${v%%:*}; flags=${v#*:}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -pthread -shared $flags dropest_amd/csrc/bgzf_api.hip dropest_amd/csrc/annotation_api.hip -o $D/libbgzf_$name.so
  echo "built $D/libbgzf_$name.so ($flags)"
done
