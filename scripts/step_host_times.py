"""Host wall time of every call of one bench step (bench.py: one_step), C2 shape: which calls the step's time sits in, and what is left
between the last kernel of a pass and the first of the next.  usage: python scripts/step_host_times.py [reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dropest_amd import capi
from dropest_amd.synth import SynthStream
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
s = SynthStream(n_reads=n, n_cells=5000, n_genes=30000, umi_len=10, stream_id=2)
dev = s.generate_device(0)
c = capi.Context(min_genes_before_merge=20, min_genes_after_merge=100)
c.push_reads_device(*dev.ptrs, dev.n, adopt=True)
names = ["set_wire", "set_prefetch", "reset_results", "set_initialized", "merge_and_filter", "prefetch_raw", "cm", "cm_raw", "filtered_cells", "merge_target_pairs"]
acc = {k: [] for k in names}
tot = []
for it in range(12):
    t = [time.perf_counter()]
    c.set_matrix_wire(True); t.append(time.perf_counter())
    c.set_raw_matrix_prefetch(0); t.append(time.perf_counter())
    c.reset_results(); t.append(time.perf_counter())
    c.set_initialized(); t.append(time.perf_counter())
    c.merge_and_filter(); t.append(time.perf_counter())
    c.prefetch_raw_matrix(form=0); t.append(time.perf_counter())
    cm = c.count_matrix_csc(filtered=True); t.append(time.perf_counter())
    raw = c.count_matrix_csc(filtered=False); t.append(time.perf_counter())
    f = c.filtered_cells(); t.append(time.perf_counter())
    m = c.merge_target_pairs(); t.append(time.perf_counter())
    if it >= 2:
        for k, a, b in zip(names, t[:-1], t[1:]):
            acc[k].append((b - a) * 1e3)
        tot.append((t[-1] - t[0]) * 1e3)
print("step %.3f ms (min %.3f)" % (np.median(tot), min(tot)))
for k in names:
    print("  %-20s median %.3f ms  min %.3f" % (k, np.median(acc[k]), min(acc[k])))
