"""cb_insert / build_keys times at C2 size under DROPEST_CB_DEBUG variants (timing experiments only; results are wrong)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
from dropest_amd import capi
from dropest_amd.synth import SynthStream
s = SynthStream(n_reads=100_000_000, n_cells=5000, n_genes=30000)
dev = s.generate_device(0)
c = capi.Context(min_genes_before_merge=20, min_genes_after_merge=100)
c.push_reads_device(*dev.ptrs, dev.n, adopt=True)
L = capi.lib()
for dbg in os.environ.get("VARIANTS", "0").split(","):
    os.environ["DROPEST_CB_DEBUG"] = dbg
    for it in range(3):
        if it == 1:
            c.set_profiling(True)
        c.reset_results()
        try:
            c._chk(L.dropest_ingest(c.h))
        except Exception as e:
            print("variant", dbg, "error", e); break
    st = c.kernel_stats()
    c.set_profiling(False)
    print("variant", dbg, {k: round(v["ms"] / v["launches"], 4) for k, v in st.items() if k.startswith("cb_")}, flush=True)
