import time, numpy as np
from dropest_amd import capi
from dropest_amd.synth import SynthStream
s = SynthStream(n_reads=100_000_000, n_cells=5000, n_genes=30000, stream_id=2)
dev = s.generate_device(0)
c = capi.Context(min_genes_before_merge=20, min_genes_after_merge=100)
c.push_reads_device(*dev.ptrs, dev.n, adopt=True)
acc = {}
def t(name, f):
    t0 = time.perf_counter(); r = f(); acc[name] = acc.get(name, 0) + time.perf_counter() - t0; return r
for it in range(12):
    if it == 2: acc.clear()
    t("reset", c.reset_results); t("init", c.set_initialized); t("merge", c.merge_and_filter)
    t("cm", lambda: c.count_matrix_csc(filtered=True)); t("raw", lambda: c.count_matrix_csc(filtered=False)); t("filt", c.filtered_cells)
print({k: round(v / 10 * 1e3, 3) for k, v in acc.items()}, "sum", round(sum(acc.values()) / 10 * 1e3, 3))
print({k: round(v["ms"] / 12, 3) for k, v in c.kernel_stats().items() if k.startswith("host:")})
