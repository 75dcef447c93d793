"""Per-kernel times of the splitter sort at C2 size under DROPEST_SS_DEBUG variants (timing experiments only)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
from dropest_amd import capi
from dropest_amd.synth import SynthStream
n = int(float(os.environ.get("N", "1e8")))
s = SynthStream(n_reads=n, n_cells=int(os.environ.get("CELLS", "5000")), n_genes=30000, umi_len=int(os.environ.get("UMI", "10")),
                stream_id=int(os.environ.get("STREAM", "2")))
dev = s.generate_device(0)
c = capi.Context(min_genes_before_merge=20, min_genes_after_merge=100)
c.push_reads_device(*dev.ptrs, dev.n, adopt=True)
for dbg in os.environ.get("VARIANTS", "0").split(","):
    os.environ["DROPEST_SS_DEBUG"] = dbg
    for it in range(3):
        if it == 1:
            c.set_profiling(True)
        c.reset_results()
        try:
            c.set_initialized()
        except Exception as e:
            print("variant", dbg, "error", e); break
    st = c.kernel_stats()
    c.set_profiling(False)
    print("variant", dbg, {k: round(v["ms"] / v["launches"], 4) for k, v in st.items() if k.startswith("ss_")}, flush=True)
