import sys, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from dropest_amd import capi
from dropest_amd.synth import SynthStream
from oracle import Oracle
import parity
rng = np.random.default_rng(424242)
widths = set()
for it in range(14):
    n = int(rng.integers(100_000, 2_500_000))
    kw = dict(n_cells=int(rng.integers(5, 400)), n_genes=int(rng.integers(50, 20000)), umi_len=int(rng.integers(5, 13)), stream_id=int(rng.integers(1, 1000)))
    s = SynthStream(n_reads=n, **kw)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    mb, ma = int(rng.integers(0, 30)), int(rng.integers(0, 120))
    t0 = time.time()
    o = parity.oracle_run(Oracle, dict(merge_kind=0, min_genes_before=mb, min_genes_after=ma, match_levels="eEBA", max_cells=-1), cb, umi, gene, aux)
    c = parity.gpu_run(dict(merge_kind=capi.MERGE_NONE, min_genes_before_merge=mb, min_genes_after_merge=ma, gene_match_levels="eEBA", max_cells=-1),
                       cb, umi, gene, aux, chunks=int(rng.integers(1, 4)))
    parity.compare(o, c)
    L = c.sort_layout()
    w = L["cell_bits"] + L["gene_bits"] + L["umi_bits"]
    widths.add((w, L["passes"], L["value_bytes"]))
    print(it, n, kw, "width", w, "passes", L["passes"], "ok %.1fs" % (time.time() - t0), flush=True)
print("layouts seen:", sorted(widths))
