"""One context reused for a different stream (dropest_clear_reads) against a fresh context on that stream: every observable equal?
(the reused context's buffers hold whatever the previous stream left)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
import numpy as np
from dropest_amd import capi
from dropest_amd.synth import SynthStream
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "dropest_amd", "data", "barcodes")
kw = dict(min_genes_before_merge=10, min_genes_after_merge=60, merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=capi.BARCODES_CONST,
          barcodes_file=os.path.join(DATA, "10x_aug_2016_split"))


def outputs(c):
    rows = c.cell_rows()
    return {"cm": [x.copy() for x in c.count_matrix_csc(filtered=True)], "raw": [x.copy() for x in c.count_matrix_csc(filtered=False)],
            "rows": {k: rows[k].copy() for k in rows.dtype.names}, "targets": np.array(c.merge_targets())}


rng = np.random.default_rng(int(os.environ.get("SEED", "5")))
for it in range(int(os.environ.get("ITERS", "4"))):
    shapes = [dict(n_reads=int(rng.integers(30_000_000, 150_000_000)), n_cells=int(rng.integers(500, 20000)), n_genes=int(rng.integers(2000, 30000)),
                   umi_len=int(rng.choice([8, 10, 12])), stream_id=int(rng.integers(1, 999)), permille_neighbour=int(rng.integers(40, 200))) for _ in range(2)]
    devs = [SynthStream(**s).generate_device(0) for s in shapes]
    c = capi.Context(**kw)
    c.push_reads_device(*devs[0].ptrs, devs[0].n, adopt=True)
    c.set_initialized(); c.merge_and_filter(); outputs(c)
    c.clear_reads()
    c.push_reads_device(*devs[1].ptrs, devs[1].n, adopt=True)
    c.set_initialized(); c.merge_and_filter()
    a = outputs(c)
    c.close()
    f = capi.Context(**kw)
    f.push_reads_device(*devs[1].ptrs, devs[1].n, adopt=True)
    f.set_initialized(); f.merge_and_filter()
    b = outputs(f)
    f.close()
    bad = [n for n in ("cm", "raw") for x, y in zip(a[n], b[n]) if not np.array_equal(x, y)] + [k for k in a["rows"] if not np.array_equal(a["rows"][k], b["rows"][k])]
    if not np.array_equal(a["targets"], b["targets"]):
        bad.append("targets")
    print(it, shapes[0]["n_reads"], "->", shapes[1]["n_reads"], "equal" if not bad else "DIFFERENT in %s" % bad, flush=True)
    for d in devs:
        d.free()
