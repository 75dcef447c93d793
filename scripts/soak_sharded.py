"""Differential soak of the sharded runner: random streams cut into ragged contiguous ranges (empty shards included) over 2-8 shards
on one GPU, with and without the barcode merges (-m / -M with the whitelist; Simple, PoissonSimple and merge-all without one), N-UMIs,
-u, UMI qualities, in the three matrix forms and with cm_raw planned
on the device or on the host -- every observable equal to ONE context over the same stream.
usage: soak_sharded.py [iterations] [seed] [max_reads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import parity
from dropest_amd import capi
from dropest_amd.multi import ShardGroup, cfg_kwargs
from dropest_amd.synth import SynthStream, inject_n

DATA = os.path.join(ROOT, "dropest_amd", "data", "barcodes")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20260929)
max_reads = int(float(sys.argv[3])) if len(sys.argv) > 3 else 3_000_000
FREE = os.environ.get("SOAK_FREE_ONLY")   # only the merges without a whitelist


def single(arrays, kw, side, qual):
    c = capi.Context(**kw)
    if side:
        c.set_side_strings(side)
    c.push_reads(*arrays)
    if qual is not None:
        c.set_umi_qualities(qual)
    c.set_initialized(); c.merge_and_filter()
    return c


def molecule_table(ctx, ql):
    """(barcode, gene, umi) -> (reads, mark, quality sums) over the real, unmerged cells of a context."""
    rows = ctx.cell_rows()
    out = {}
    for cell in np.flatnonzero(rows["is_real"].astype(bool) & ~rows["is_merged"].astype(bool)):
        g, u, r, m = ctx.cell_molecules(int(cell))
        q = ctx.cell_molecule_qualities(int(cell), len(g)) if ql and len(g) else None
        b = int(rows["barcode"][cell])
        for i in range(len(g)):
            out[(b, int(g[i]), int(u[i]))] = (int(r[i]), int(m[i]), tuple(int(x) for x in q[i]) if q is not None else ())
    return out


fails = 0
for it in range(iters):
    n = int(rng.integers(20_000, max_reads))
    merge = int(rng.integers(0, 6)) if FREE is None else int(rng.choice([3, 4, 5]))   # 0 none, 1 -m, 2 -M (whitelist); 3 Simple, 4 PoissonSimple, 5 all
    free = merge >= 3
    indrop = merge and rng.random() < 0.5
    tiny = free and rng.random() < 0.4                  # few UMIs and genes: near-ties, replays with the reference's containers
    kw = dict(n_reads=n, n_cells=int(rng.integers(5, 300)), n_genes=int(rng.integers(30, 20000)), umi_len=8 if indrop else int(rng.integers(6, 13)),
              stream_id=int(rng.integers(1, 10000)), permille_neighbour=int(rng.integers(0, 300)) if merge else 0)
    if tiny:
        kw.update(n_reads=min(n, 400_000), n_cells=int(rng.integers(5, 40)), n_genes=int(rng.integers(5, 80)), umi_len=int(rng.integers(3, 6)))
        n = kw["n_reads"]
    if merge:
        kw["whitelist"] = "indrop_v3" if indrop else "10x_aug_2016_split"
    cfg = {"min_before": int(rng.integers(0, 12)), "min_after": int(rng.integers(0, 60))}
    if merge and not free:
        cfg["merge"] = {"barcodes_kind": capi.BARCODES_CONST, "barcodes_file": os.path.join(DATA, kw["whitelist"])}
    ckw = cfg_kwargs(cfg)
    if merge == 2:
        ckw.update(merge_kind=capi.MERGE_POISSON_REAL)
    if free:
        ckw.update(merge_kind={3: capi.MERGE_SIMPLE, 4: capi.MERGE_POISSON_SIMPLE, 5: capi.MERGE_ALL}[merge],
                   max_cb_merge_edit_distance=int(rng.integers(1, 18 if tiny else 4)), min_merge_fraction=float(rng.choice([0.0, 0.05, 0.2])),
                   max_real_merge_prob=float(rng.choice([1e-7, 1e-3, 0.3, 0.9])))
    if rng.random() < 0.15:
        ckw.update(umi_merge_kind=capi.UMI_MERGE_DIRECTIONAL, max_umi_merge_edit_distance=1, umi_merge_multiplier=2.0)
    cb, umi, gene, aux = parity.canonical_stream(*SynthStream(**kw).generate_host())
    side = ()
    if rng.random() < 0.3:
        umi, side = inject_n(umi, gene, float(10 ** rng.uniform(-4, -2)), int(rng.integers(1, 100)), kw["umi_len"])
    ql = int(rng.integers(4, 13)) if rng.random() < 0.4 and n < 3_000_000 else 0   # (the molecule tables are compared in Python)
    qual = rng.integers(33, 75, (n, ql)).astype(np.uint8) if ql else None
    world = int(rng.choice([2, 3, 4, 5, 8]))
    cuts = np.sort(rng.integers(0, n + 1, world - 1)) if rng.random() < 0.5 else np.array([n * i // world for i in range(1, world)])
    bounds = [0] + [int(x) for x in cuts] + [n]
    opts = {"byte_matrix": int(rng.random() < 0.7), "narrow_matrix": int(rng.random() < 0.6), "raw_on_device": int(rng.random() < 0.8),
            "packed_exchange": int(rng.random() < 0.8), "byte_list_cap": int(rng.choice([0, 0, 0, 16, 4096])),
            "slots_matrix": int(rng.random() < 0.7), "exchange_chunks": int(rng.choice([1, 1, 2, 4, 7]))}   # (round 5: the slots end point, the all-to-all in chunks)
    tag = "it %d: n %d world %d merge %d N-UMIs %d qual %d bounds %s opts %s %s" % (it, n, world, merge, len(side), ql, bounds, opts, kw)
    t0 = time.time()
    if it < int(os.environ.get("SOAK_START", "0")):   # (every random draw of the case is behind us: the cases after it stay what they were)
        continue
    if os.environ.get("SOAK_VERBOSE"):
        print("run  %s" % tag, flush=True)
    try:
        g = ShardGroup([0] * world, **ckw)
        for i, s in enumerate(g.shards):
            for k, v in opts.items():
                s.set_option(k, v)
            if side:
                s.set_side_strings(side)
            lo, hi = bounds[i], bounds[i + 1]
            s.set_reads(capi.DeviceArrays.from_host(0, cb[lo:hi], umi[lo:hi], gene[lo:hi], aux[lo:hi]), lo)
            if ql:
                s.set_umi_qualities(qual[lo:hi])
        for _ in range(2):
            g.step()
        s0 = g.shards[0]
        got = {"cm": [x.copy() for x in s0.matrix(True)], "raw": [x.copy() for x in s0.matrix(False)], "merged": s0.merged_barcodes()}
        c = single((cb, umi, gene, aux), ckw, side, qual)
        rows = c.cell_rows()
        for filt, name in ((True, "cm"), (False, "raw")):
            p, i_, x = c.count_matrix_csc(filtered=filt)
            gp, gi, gx, gb = got[name]
            assert np.array_equal(gp.astype(np.uint64), p.astype(np.uint64)), name + " colptr"
            assert np.array_equal(gi, i_) and np.array_equal(gx, x), name + " entries"
        assert [int(b) for b in got["cm"][3]] == [int(rows["barcode"][int(k)]) for k in c.filtered_cells()], "cm columns"
        assert [int(b) for b in got["raw"][3]] == [int(b) for b in rows["barcode"][rows["is_real"].astype(bool)]], "raw columns"
        mt = c.merge_targets()
        src = np.flatnonzero(mt != np.arange(len(mt)))
        want = {int(rows["barcode"][k]): int(rows["barcode"][int(mt[k])]) for k in src}
        assert dict(zip((int(b) for b in got["merged"][0]), (int(b) for b in got["merged"][1]))) == want, "merged pairs"
        if ql:       # the molecules with their quality sums, shard by shard, against the one context
            have = {}
            for s in g.shards:
                have.update(molecule_table(s.ctx, ql))
            assert have == molecule_table(c, ql), "molecules / quality sums"
        g.close(); c.close()
        print("ok   %s  cm %d raw %d merged %d  %.1fs" % (tag, len(got["cm"][1]), len(got["raw"][1]), len(want), time.time() - t0), flush=True)
    except (AssertionError, capi.DropestError) as e:
        if isinstance(e, capi.DropestError) and ("not supported" in str(e).lower() or "unsupported" in str(e).lower() or "sharded runs support" in str(e)):
            print("skip %s  (%s)" % (tag, str(e)[:120]), flush=True)
            continue
        fails += 1
        print("FAIL %s  %s: %s" % (tag, type(e).__name__, e), flush=True)
print("failures:", fails)
sys.exit(1 if fails else 0)
