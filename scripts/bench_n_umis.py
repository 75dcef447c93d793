"""Developer tool: cost of the N-UMI merge (MergeUMIsStrategySimple) at a given N rate, host-generated stream."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import parity
from dropest_amd import capi
from dropest_amd.synth import SynthStream, inject_n
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
rate = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
s = SynthStream(n_reads=n, n_cells=2000, n_genes=20000, umi_len=10)
cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
t0 = time.time()
umi, side = inject_n(umi, gene, rate, 7, 10)
print("N reads:", len(side), "inject %.1fs" % (time.time() - t0))
for kind in (capi.UMI_MERGE_SIMPLE, capi.UMI_MERGE_DIRECTIONAL):
    c = capi.Context(min_genes_before_merge=20, min_genes_after_merge=100, umi_merge_kind=kind)
    c.set_side_strings(side); c.push_reads(cb, umi, gene, aux)
    c.set_profiling(True)
    c.set_initialized()
    t0 = time.time(); c.merge_and_filter(); dt = time.time() - t0
    st = c.kernel_stats()
    print("kind", kind, "merge_and_filter %.1f ms" % (dt * 1e3), {k: round(v["ms"], 2) for k, v in st.items() if k.startswith("host:") and "umi" in k or "count:" in k})
