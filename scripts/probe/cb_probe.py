"""Per-kernel times of the barcode-table stage at C2 size under CB_EXP variants (timing experiments only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
from dropest_amd import capi
from dropest_amd.synth import SynthStream
n = int(float(os.environ.get("N", "1e8")))
s = SynthStream(n_reads=n, n_cells=int(os.environ.get("CELLS", "5000")), n_genes=30000, umi_len=int(os.environ.get("UMI", "10")))
dev = s.generate_device(0)
c = capi.Context(min_genes_before_merge=20, min_genes_after_merge=100)
c.push_reads_device(*dev.ptrs, dev.n, adopt=True)
for v in os.environ.get("VARIANTS", "0").split(","):
    os.environ["CB_EXP"] = v
    for it in range(3):
        if it == 1:
            c.set_profiling(True)
        c.reset_results()
        try:
            c.ingest() if os.environ.get("ONLY_INGEST") else c.set_initialized()
        except Exception as e:
            print("variant", v, "error", str(e)[:100]); break
    st = c.kernel_stats()
    c.set_profiling(False)
    print("variant", v, {k: round(x["ms"] / x["launches"], 4) for k, x in st.items() if k.startswith("cb_") or k == "build_keys"}, flush=True)
