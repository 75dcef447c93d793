"""Per-kernel roofline table of one bench.py run: algorithmic bytes and HIP-event time from the bench line's
`kernels_ms_per_step` is not enough (no bytes), so this tool re-runs one profiled step itself through the C-ABI and
joins it with the PMC traffic summary (profiles/<tag>_pmc_summary.csv) when present.

    python scripts/roofline_table.py profiles/r01d_pmc_summary.csv > profiles/r01d_roofline_by_kernel.md"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dropest_amd import capi                          # noqa: E402
from dropest_amd.capi import Context                  # noqa: E402
from dropest_amd.synth import SynthStream             # noqa: E402

PEAK = 8000.0   # GB/s, MI355X HBM3E
n = 100_000_000
stream = SynthStream(n_reads=n, n_cells=5000, n_genes=30000, cb_len=16, umi_len=10, stream_id=2)
dev = stream.generate_device(0, first=0, n=n)
ctx = Context(device=0, merge_kind=capi.MERGE_NONE, min_genes_before_merge=20, min_genes_after_merge=100)
ctx.push_reads_device(*dev.ptrs, dev.n, adopt=True)


def step():
    ctx.reset_results(); ctx.set_initialized(); ctx.merge_and_filter()
    ctx.count_matrix_csc(filtered=True); ctx.count_matrix_csc(filtered=False)


for _ in range(2):
    step()
ctx.set_profiling(True)
K = 5
for _ in range(K):
    step()
stats = ctx.kernel_stats()
pmc = {}
if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):
    for row in csv.DictReader(l for l in open(sys.argv[1]) if not l.startswith("#")):
        pmc[row["kernel"]] = float(row["hbm_bytes_per_launch"])
# kernel-stat name -> substring of the rocprof kernel name
alias = {"ss_local:keys": "ss_local_kernel<0", "ss_local:big": "ss_local_big_kernel<", "ss_scatter:L1:keys": "ss_scatter_res_l1_kernel<0", "ss_scatter:L2:keys": "ss_scatter_res_l2_kernel<0",
         "build_keys+L1": "build_keys_scatter_kernel", "ss_compact:cell_gene": "ss_compact_cg_kernel", "emit_matrix:cm": "emit_matrix_kernel<2>", "emit_matrix:cm_raw": "emit_matrix_bytes_short_kernel",
         "ss_hist:L1": "ss_hist_l1_kernel", "ss_hist:L2": "ss_hist_l2_kernel", "ss_compact": "ss_compact_kernel", "ss_sample": "ss_sample_", "cb_sample": "cb_sample_distinct_kernel",
         "rs_scatter:keys": "rs_scatter_kernel_t<512, 8, false, 0, 8>", "rs_hist": "rs_hist_kernel", "cb_insert": "cb_insert_",
         "build_keys": "build_keys_kernel", "seg_reduce:molecules": "seg_reduce_kernel<ReadsToMoleculesX<0>",
         "seg_reduce:cell_gene": "seg_reduce_kernel<MoleculesToCellGeneX>", "seg_reduce:cells": "seg_reduce_kernel<CellGeneToCells>",
         "seg_count:molecules": "seg_count_kernel<ReadsToMoleculesX<0>", "seg_count:cell_gene": "seg_count_kernel<MoleculesToCellGeneX>",
         "cb_compact_slots": "cb_compact_slots_kernel", "cb_assign_ids": "cb_assign_sorted_kernel"}
print("# Kernels of one C2 pass (1e8 reads, 1 x MI355X): algorithmic bytes, HIP-event time, achieved vs the 8 TB/s HBM peak")
print()
print("| kernel | launches / pass | ms / pass | algorithmic GB / pass | achieved GB/s | % of HBM peak | PMC HBM bytes / algorithmic |")
print("|---|---|---|---|---|---|---|")
rows = [(k, v) for k, v in stats.items() if not k.startswith("host:") and not k.startswith("count:") and v["ms"] > 0]
rows.sort(key=lambda kv: -kv[1]["ms"])
tot_ms = 0.0
for k, v in rows:
    ms, gb, launches = v["ms"] / K, v["bytes"] / K / 1e9, v["launches"] / K
    tot_ms += ms
    amp = ""
    key = alias.get(k)
    if key:
        hit = sorted((b for name, b in pmc.items() if name.startswith(key)), reverse=True)   # several grid sizes: the main launches
        if hit and v["bytes"]:
            amp = "%.2f" % (hit[0] * v["launches"] / v["bytes"])
    print("| `%s` | %.0f | %.3f | %.3f | %.0f | %.1f | %s |" % (k, launches, ms, gb, gb / ms * 1e3 if ms else 0, gb / ms * 1e3 / PEAK * 100 if ms else 0, amp))
print()
print("Sum of kernel time: %.2f ms per pass (the pass ends with the two matrices crossing PCIe in the byte form, ~1.5 ms, and their decode on host threads)." % tot_ms)
