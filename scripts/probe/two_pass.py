"""Same context, same reads, two passes (dropest_reset_results in between): every observable equal?  (buffers keep the first pass's content)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
import numpy as np
from dropest_amd import capi
from dropest_amd.synth import SynthStream
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "dropest_amd", "data", "barcodes")
shape = {'n_reads': int(float(os.environ.get("N", "145507637"))), 'n_cells': 17642, 'n_genes': 8184, 'umi_len': 12, 'stream_id': 370, 'permille_neighbour': 97}
dev = SynthStream(**shape).generate_device(0)
kw = dict(min_genes_before_merge=10, min_genes_after_merge=60, merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=capi.BARCODES_CONST,
          barcodes_file=os.path.join(DATA, "10x_aug_2016_split"))
c = capi.Context(**kw)
c.push_reads_device(*dev.ptrs, dev.n, adopt=True)
outs = []
for p in range(3):
    c.reset_results()
    c.set_initialized(); c.merge_and_filter()
    rows = c.cell_rows()
    outs.append({"cm": [x.copy() for x in c.count_matrix_csc(filtered=True)], "raw": [x.copy() for x in c.count_matrix_csc(filtered=False)],
                 "rows": {k: rows[k].copy() for k in rows.dtype.names}, "targets": np.array(c.merge_targets())})
    if p:
        a, b = outs[0], outs[p]
        bad = [n for n in ("cm", "raw") for x, y in zip(a[n], b[n]) if not np.array_equal(x, y)] + [k for k in a["rows"] if not np.array_equal(a["rows"][k], b["rows"][k])]
        if not np.array_equal(a["targets"], b["targets"]):
            bad.append("targets")
        print("pass", p, "vs pass 0:", "equal" if not bad else "DIFFERENT in %s" % bad, flush=True)
