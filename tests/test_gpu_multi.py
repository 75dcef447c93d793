"""The sharded runner (csrc/shard_run.h through dropest_shard_*): two / three shards -- all on ONE GPU, which an in-process
group allows -- must assemble exactly what a single context computes over the whole stream: both matrices, their column
barcodes, the merged barcodes.  Whitelist merges whose targets live on other shards, N-UMIs (one global rand() sequence,
global first occurrences), barcodes of several lengths, -C; and one shard over RCCL (a world of one process)."""
import os

import numpy as np
import pytest

from dropest_amd import capi
from dropest_amd.multi import ShardGroup, ShardedRun, cfg_kwargs, widen_shard_matrix
from dropest_amd.synth import SynthStream, inject_n

import parity

pytestmark = pytest.mark.gpu

# soak runs: DROPEST_MULTI_SCALE=<k> multiplies reads and cells of every case of this file (default 1 = the committed cases)
SCALE = int(os.environ.get("DROPEST_MULTI_SCALE", "1"))
CFG = {"min_before": 10, "min_after": 30}
STREAM = dict(n_reads=400_000 * SCALE, n_cells=60 * SCALE, n_genes=3000)
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dropest_amd", "data", "barcodes")
MERGE_CASES = {
    # name: (stream parameters, whitelist file, barcodes kind, thresholds)
    "10x": (dict(n_reads=300_000 * SCALE, n_cells=40 * SCALE, n_genes=2000, umi_len=12, permille_neighbour=150), "10x_aug_2016_split",
            capi.BARCODES_CONST, {"min_before": 3, "min_after": 20}),
    "indrop": (dict(n_reads=300_000 * SCALE, n_cells=40 * SCALE, n_genes=2000, umi_len=8, permille_neighbour=150, whitelist="indrop_v3"),
               "indrop_v3", capi.BARCODES_CONST, {"min_before": 3, "min_after": 20}),
}


def run_group(world, arrays, cfg_kw, side=(), steps=2):
    """arrays = (cb, umi, gene, aux) of the whole stream (host): shard i gets the i-th contiguous range."""
    n = len(arrays[0])
    bounds = [n * i // world for i in range(world + 1)]
    g = ShardGroup([0] * world, **cfg_kw)
    for i, s in enumerate(g.shards):
        if side:
            s.set_side_strings(side)
        s.set_reads(capi.DeviceArrays.from_host(0, *[a[bounds[i]:bounds[i + 1]] for a in arrays]), bounds[i])
    for _ in range(steps):                      # a second step exercises buffer reuse
        g.step()
    s0 = g.shards[0]
    out = {"cm": [x.copy() for x in s0.matrix(True)], "raw": [x.copy() for x in s0.matrix(False)], "merged": s0.merged_barcodes(),
           "phases": s0.phase_stats()}
    g.close()
    return out


def single(arrays, cfg_kw, side=()):
    c = capi.Context(**cfg_kw)
    if side:
        c.set_side_strings(side)
    c.push_reads(*arrays)
    c.set_initialized(); c.merge_and_filter()
    return c


def check(got, c):
    rows = c.cell_rows()
    for filt, name in ((True, "cm"), (False, "raw")):
        p, i, x = c.count_matrix_csc(filtered=filt)
        gp, gi, gx, gb = got[name]
        assert np.array_equal(gp.astype(np.uint64), p.astype(np.uint64)), name
        assert np.array_equal(gi, i) and np.array_equal(gx, x), name
    assert [int(b) for b in got["cm"][3]] == [int(rows["barcode"][int(k)]) for k in c.filtered_cells()]
    assert [int(b) for b in got["raw"][3]] == [int(b) for b in rows["barcode"][rows["is_real"].astype(bool)]]
    mt = c.merge_targets()
    src = np.flatnonzero(mt != np.arange(len(mt)))
    want = {int(rows["barcode"][k]): int(rows["barcode"][int(mt[k])]) for k in src}
    have = dict(zip((int(b) for b in got["merged"][0]), (int(b) for b in got["merged"][1])))
    assert have == want
    return want


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shards_on_one_gpu_match_single_context(world):
    arrays = parity.canonical_stream(*SynthStream(**STREAM).generate_host())
    kw = cfg_kwargs(CFG)
    got = run_group(world, arrays, kw)
    check(got, single(arrays, kw))
    assert len(got["cm"][3]) > 20
    if world > 1:
        assert got["phases"]["all_to_all"]["bytes"] > 0          # reads really moved between the shards


@pytest.mark.parametrize("what", ["chromosome", "umi"])
def test_a_field_wider_than_the_sampled_reads_suggest(what):
    """The widths of the 12-byte exchange record come from every 64th row of 512 reads; ONE read elsewhere with a chromosome id beyond them
    is noticed by the scatter and the partition is repeated with exact widths on every shard (same results; the phase shows in the
    statistics); afterwards the shards keep exact widths.  A longer UMI still fits (the UMI takes every bit the barcode leaves)."""
    arrays = [a.copy() for a in parity.canonical_stream(*SynthStream(**STREAM).generate_host())]
    at = 512 * 3 + 17                                      # a read of an unsampled row of the first shard's range
    if what == "chromosome":
        arrays[3][at] = (int(arrays[3][at]) & 0xFFFF0000) | 0x1FFF
    else:
        arrays[1][at] = int(capi.pack_seq("ACGTACGTACGTACG"))
    kw = cfg_kwargs(CFG)
    got = run_group(2, arrays, kw)
    check(got, single(arrays, kw))
    assert ("partition:exact_again" in got["phases"]) == (what == "chromosome")
    if what == "chromosome":
        assert got["phases"]["partition:exact_again"]["steps"] == 1     # the first step repeated its partition, the second knew better


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", sorted(MERGE_CASES))
def test_sharded_whitelist_merge_matches_single_context(case, world):
    """-m with a whitelist over 2 / 3 shards: merge targets on other shards, molecule rows moving between shards."""
    kw, wl, kind, cfg = MERGE_CASES[case]
    arrays = parity.canonical_stream(*SynthStream(**kw).generate_host())
    ckw = cfg_kwargs(dict(cfg, merge={"barcodes_kind": kind, "barcodes_file": os.path.join(DATA, wl)}))
    got = run_group(world, arrays, ckw)
    c = single(arrays, ckw)
    want = check(got, c)
    assert len(want) > 20 and int(c.cell_rows()["is_excluded"].sum()) > 0
    owner = lambda b: capi.lib().dropest_owner_of(int(b), world)           # noqa: E731
    assert sum(owner(a) != owner(b) for a, b in want.items()) > 5          # the merge really crossed shards


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", sorted(MERGE_CASES))
def test_sharded_poisson_whitelist_merge_matches_single_context(case, world):
    """-M with a whitelist (PoissonRealBarcodesMergeStrategy) over 2 / 3 shards: the UMI distribution and the largest gene are those of ALL
    shards (histograms added, every shard builds the same estimator tables), the expected intersection of a pair is computed where the
    candidate lives from the base's shipped rows, the decision where the base lives.  Same targets, same matrices as one context."""
    kw, wl, kind, cfg = MERGE_CASES[case]
    arrays = parity.canonical_stream(*SynthStream(**kw).generate_host())
    ckw = cfg_kwargs(dict(cfg, merge={"barcodes_kind": kind, "barcodes_file": os.path.join(DATA, wl)}))
    ckw.update(merge_kind=capi.MERGE_POISSON_REAL)
    got = run_group(world, arrays, ckw)
    c = single(arrays, ckw)
    want = check(got, c)
    assert len(want) > 20
    owner = lambda b: capi.lib().dropest_owner_of(int(b), world)           # noqa: E731
    assert sum(owner(a) != owner(b) for a, b in want.items()) > 5          # the merge really crossed shards
    # not the plain whitelist merge's answer by accident: -M decides differently on this stream
    plain = single(arrays, dict(ckw, merge_kind=capi.MERGE_REAL_BARCODES))
    assert not np.array_equal(plain.merge_targets(), c.merge_targets())


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("poisson", [False, True])
@pytest.mark.parametrize("n_parts", [5, 8])
def test_sharded_whitelists_of_more_than_four_parts(n_parts, poisson, world, tmp_path):
    """Const-length whitelists of any number of lines (ConstLengthBarcodesParser.cpp:50-68): beyond four parts the neighbour search runs
    on the host, every shard for its own bases against the all-gathered cell list (refused in sharded runs before round 4)."""
    from test_gpu_stress import random_whitelist_case
    crossed = 0
    for seed in range(3):
        cb, umi, gene, aux, side, _, gkw = random_whitelist_case(100 * n_parts + seed, poisson, tmp_path, force_parts=n_parts)
        arrays = (cb, umi, gene, aux)
        got = run_group(world, arrays, gkw, side=side)
        want = check(got, single(arrays, gkw, side=side))
        owner = lambda b: capi.lib().dropest_owner_of(int(b), world)       # noqa: E731
        crossed += sum(owner(a) != owner(b) for a, b in want.items())
    if n_parts == 5 and not poisson:
        assert crossed > 3                                                # merges really crossed shards (the other cases: a handful of merges each)


FREE_CASES = {
    # name: (stream, configuration): the parity cases of tests/test_gpu_parity.py for the same strategies on one context
    "plain": (dict(n_reads=150_000 * SCALE, n_cells=25 * SCALE, n_genes=1200, umi_len=8, permille_neighbour=150),
              dict(max_cb_merge_edit_distance=2, min_merge_fraction=0.2, min_genes_before_merge=3, min_genes_after_merge=10)),
    "collisions": (dict(n_reads=150_000 * SCALE, n_cells=25 * SCALE, n_genes=1200, umi_len=5, permille_neighbour=150),
                   dict(max_cb_merge_edit_distance=3, min_merge_fraction=0.05, min_genes_before_merge=3, min_genes_after_merge=10)),
    "ties": (dict(n_reads=60_000 * SCALE, n_cells=12 * SCALE, n_genes=40, umi_len=3, permille_neighbour=250),
             dict(max_cb_merge_edit_distance=17, min_merge_fraction=0.0, min_genes_before_merge=1, min_genes_after_merge=1)),
}


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("case", sorted(FREE_CASES))
def test_sharded_simple_merge_matches_single_context(case, world):
    """-m without a whitelist (SimpleMergeStrategy.cpp:16-108) over shards: the UMI-gene index sharded by hash(UMI-gene), partial pair
    counts routed to the owners of the bases, near-ties replayed with the reference's containers over GLOBAL cell indices and the global
    UMI order.  Same targets, same matrices as one context."""
    kw, cfg = FREE_CASES[case]
    arrays = parity.canonical_stream(*SynthStream(**kw).generate_host())
    ckw = dict(cfg, merge_kind=capi.MERGE_SIMPLE)
    got = run_group(world, arrays, ckw)
    c = single(arrays, ckw)
    want = check(got, c)
    assert len(want) > (300 if case != "ties" else 20)
    owner = lambda b: capi.lib().dropest_owner_of(int(b), world)           # noqa: E731
    assert sum(owner(a) != owner(b) for a, b in want.items()) > 5          # the merge really crossed shards
    if case == "ties":
        assert "cbm:replay" in got["phases"]                               # the replay ran across the shards


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("max_ed", [1, 2, 4])
def test_sharded_merge_all_matches_single_context(max_ed, world):
    """merge_type = all (MergeAllMergeStrategy.h:16-50) over shards: every shard decides its own cells against the barcodes of all."""
    s = SynthStream(n_reads=150_000 * SCALE, whitelist="10x_aug_2016_split", n_cells=30 * SCALE, n_genes=1500, umi_len=10, permille_neighbour=150)
    arrays = parity.canonical_stream(*s.generate_host())
    ckw = dict(merge_kind=capi.MERGE_ALL, max_cb_merge_edit_distance=max_ed, min_genes_before_merge=3, min_genes_after_merge=10)
    got = run_group(world, arrays, ckw)
    want = check(got, single(arrays, ckw))
    assert len(want) > 20
    owner = lambda b: capi.lib().dropest_owner_of(int(b), world)           # noqa: E731
    assert sum(owner(a) != owner(b) for a, b in want.items()) > 5


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("max_ed,p_real,umi_len", [(2, 1e-7, 10), (1, 1e-3, 8), (3, 0.5, 8)])
def test_sharded_poisson_simple_merge_matches_single_context(max_ed, p_real, umi_len, world):
    """-M without a whitelist (PoissonSimpleMergeStrategy.cpp:15-43) over shards: pairs from the sharded UMI-gene index, the estimator's
    tables from the UMI distribution of all shards, every pair's expected intersection computed where the neighbour lives."""
    s = SynthStream(n_reads=150_000 * SCALE, whitelist="10x_aug_2016_split", n_cells=30 * SCALE, n_genes=1500, umi_len=umi_len, permille_neighbour=150)
    arrays = parity.canonical_stream(*s.generate_host())
    ckw = dict(merge_kind=capi.MERGE_POISSON_SIMPLE, max_cb_merge_edit_distance=max_ed, max_real_merge_prob=p_real, min_genes_before_merge=3,
               min_genes_after_merge=10)
    got = run_group(world, arrays, ckw)
    want = check(got, single(arrays, ckw))
    assert len(want) > 20
    owner = lambda b: capi.lib().dropest_owner_of(int(b), world)           # noqa: E731
    assert sum(owner(a) != owner(b) for a, b in want.items()) > 5


def test_sharded_poisson_simple_merge_with_ties():
    """Few UMIs and genes: equal probabilities, the first neighbour in the reference's map order wins -- replayed across three shards."""
    kw, _ = FREE_CASES["ties"]
    arrays = parity.canonical_stream(*SynthStream(**kw).generate_host())
    ckw = dict(merge_kind=capi.MERGE_POISSON_SIMPLE, max_cb_merge_edit_distance=17, max_real_merge_prob=0.9, min_genes_before_merge=1,
               min_genes_after_merge=1)
    got = run_group(3, arrays, ckw)
    check(got, single(arrays, ckw))


def _barcodes_with_n(cb, seed, n_chosen=25, min_reads=150):
    """40 % of the reads of some busy barcodes get one N-variant of it (an escaped code): a smaller cell next to the original one."""
    rng = np.random.default_rng(seed)
    codes, counts = np.unique(cb, return_counts=True)
    busy = codes[counts >= min_reads]
    chosen = rng.choice(busy, min(n_chosen, len(busy)), replace=False)
    side = []
    cb = cb.copy()
    for c in chosen:
        s = capi.unpack_code(int(c))
        p = int(rng.integers(0, len(s)))
        side.append(s[:p] + "N" + s[p + 1:])
        where = np.flatnonzero(cb == c)
        cb[where[:int(len(where) * 0.4)]] = capi.ESCAPE | (len(side) - 1)
    return cb, side


@pytest.mark.parametrize("kind", ["all", "simple"])
def test_sharded_free_merges_with_n_in_barcodes(kind):
    """Barcodes with N (escaped codes): the host comparison of merge_type = all, the host edit distance of the simple merge; two shards."""
    s = SynthStream(n_reads=120_000, whitelist="10x_aug_2016_split", n_cells=20, n_genes=800, umi_len=8, permille_neighbour=150)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    cb, side = _barcodes_with_n(cb, 4, n_chosen=12)
    arrays = parity.canonical_stream(cb, umi, gene, aux)
    ckw = dict(merge_kind=capi.MERGE_ALL if kind == "all" else capi.MERGE_SIMPLE, max_cb_merge_edit_distance=2, min_genes_before_merge=3,
               min_genes_after_merge=10)
    got = run_group(2, arrays, ckw, side=side)
    want = check(got, single(arrays, ckw, side=side))
    assert len(want) > 10


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("rate,n_reads", [(1e-2, 120_000), (1e-3, 600_000)])
def test_n_umis_across_shards(world, rate, n_reads):
    """UMIs with N are the reference's DEFAULT UMI merge (MergeUMIsStrategySimple.cpp:21-102): random fills follow one
    srand(42) sequence in global cell order, ties go to the UMI seen first in the WHOLE stream."""
    s = SynthStream(n_reads=n_reads * SCALE, n_cells=40 * SCALE, n_genes=1500, umi_len=8)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    umi, side = inject_n(umi, gene, rate, 7, 8)
    assert len(side) > 20
    kw = cfg_kwargs({"min_before": 10, "min_after": 20})
    got = run_group(world, (cb, umi, gene, aux), kw, side)
    c = single((cb, umi, gene, aux), kw, side)
    check(got, c)
    # (the single context itself is pinned on the oracle for these streams: test_gpu_parity.py::test_n_umis_synthetic)


@pytest.mark.parametrize("case", ["c2", "c4", "c4_simple", "c4_poisson_simple"])
def test_two_shards_at_2e7_reads_match_single_context(case):
    """Past the size thresholds of one context (sampled table, hot list, planned key layout, splitter sort) the shards still run
    the exact-statistics ingest and agree on the key fields: 2e7 reads over two shards against one context, both matrices and the
    merged barcodes."""
    if case == "c2":
        s = SynthStream(n_reads=20_000_000, n_cells=1000, n_genes=30000)
        kw = cfg_kwargs({"min_before": 20, "min_after": 100})
    else:
        s = SynthStream(n_reads=12_000_000, n_cells=800, n_genes=20000, umi_len=8, whitelist="indrop_v3", permille_neighbour=100)
        kw = cfg_kwargs({"min_before": 10, "min_after": 50, "merge": {"barcodes_kind": capi.BARCODES_CONST, "barcodes_file": os.path.join(DATA, "indrop_v3")}})
        if case != "c4":   # the same stream merged without the whitelist: 6e4 cells in the UMI-gene index, 5e6 molecules through the all-to-all
            kw = dict(merge_kind=capi.MERGE_SIMPLE if case == "c4_simple" else capi.MERGE_POISSON_SIMPLE, max_cb_merge_edit_distance=2,
                      min_merge_fraction=0.2, max_real_merge_prob=1e-7, min_genes_before_merge=10, min_genes_after_merge=50)
    dev = s.generate_device(0)
    arrays = tuple(a.copy() for a in dev.to_host())
    dev.free()
    got = run_group(2, arrays, kw, steps=1)
    c = single(arrays, kw)
    assert c.sort_layout()["sort"] == "splitter"
    want = check(got, c)
    assert len(got["cm"][3]) > 500 and (case == "c2" or len(want) > 1000)
    if case.startswith("c4_"):
        print(case, "one context: cb_merge", {k: round(v["ms"], 2) for k, v in c.kernel_stats().items() if k.startswith("host:cb_merge")},
              "two shards:", {k: round(v["ms"], 2) for k, v in got["phases"].items() if k.startswith("cbm:") or k == "cb_merge"})


def test_eight_shards_whitelist_merge_n_umis_and_directional():
    """The shard count of one node (C5's form, scaled down): whitelist merge with N-UMIs, and -u, over 8 shards on one GPU."""
    s = SynthStream(n_reads=400_000 * SCALE, n_cells=80 * SCALE, n_genes=1500, umi_len=8, permille_neighbour=150)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    umi, side = inject_n(umi, gene, 3e-3, 21, 8)
    wl = {"barcodes_kind": capi.BARCODES_CONST, "barcodes_file": os.path.join(DATA, "10x_aug_2016_split")}
    ckw = cfg_kwargs({"min_before": 3, "min_after": 10, "merge": wl})
    want = check(run_group(8, (cb, umi, gene, aux), ckw, side, steps=1), single((cb, umi, gene, aux), ckw, side))
    assert len(want) > 30
    dkw = dict(ckw, umi_merge_kind=capi.UMI_MERGE_DIRECTIONAL, max_umi_merge_edit_distance=1, umi_merge_multiplier=2.0)
    check(run_group(8, (cb, umi, gene, aux), dkw, side, steps=1), single((cb, umi, gene, aux), dkw, side))


@pytest.mark.parametrize("world", [2, 3])
def test_barcodes_with_n_in_a_sharded_whitelist_merge(world):
    """Cells whose barcode carries an N (an escaped code; Tools::edit_distance treats N as a wildcard, UtilFunctions.cpp:48) as
    BASES of a whitelist merge whose targets live on other shards (refused in rounds 1-2)."""
    stream_kw, wl, kind, cfg = MERGE_CASES["10x"]
    cb, umi, gene, aux = parity.canonical_stream(*SynthStream(**stream_kw).generate_host())
    rng = np.random.default_rng(3)
    codes, counts = np.unique(cb, return_counts=True)
    busy = codes[counts >= 150]
    chosen = rng.choice(busy, min(25, len(busy)), replace=False)
    side = []
    cb = cb.copy()
    for c in chosen:            # 40 % of the barcode's reads get one N-variant of it: a smaller cell next to the original one
        s = capi.unpack_code(int(c))
        p = int(rng.integers(0, len(s)))
        side.append(s[:p] + "N" + s[p + 1:])
        where = np.flatnonzero(cb == c)
        cb[where[:int(len(where) * 0.4)]] = capi.ESCAPE | (len(side) - 1)
    kw = dict(cfg_kwargs(dict(cfg, merge={"barcodes_kind": kind, "barcodes_file": os.path.join(DATA, wl), "min_merge_fraction": 0.0})))
    got = run_group(world, (cb, umi, gene, aux), kw, side)
    c = single((cb, umi, gene, aux), kw, side)
    want = check(got, c)
    rows = c.cell_rows()
    n_escaped_merged = sum(1 for b in want if b & capi.ESCAPE)
    assert n_escaped_merged >= 5, n_escaped_merged           # N-barcodes really were merged into whitelist cells


def test_n_umis_with_whitelist_merge_across_shards():
    s = SynthStream(n_reads=150_000 * SCALE, n_cells=25 * SCALE, n_genes=1200, umi_len=8, permille_neighbour=150)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    umi, side = inject_n(umi, gene, 5e-3, 11, 8)
    ckw = cfg_kwargs({"min_before": 3, "min_after": 10, "merge": {"barcodes_kind": capi.BARCODES_CONST,
                                                                  "barcodes_file": os.path.join(DATA, "10x_aug_2016_split")}})
    got = run_group(3, (cb, umi, gene, aux), ckw, side)
    want = check(got, single((cb, umi, gene, aux), ckw, side))
    assert len(want) > 10


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("n_rate", [0.0, 2e-2])
def test_directional_umi_correction_across_shards(world, n_rate):
    """-u (MergeUMIsStrategyDirectional.cpp:18-116) over shards: the UMI index order is the order of first appearance in the WHOLE
    stream (all shards' tables reduced to one rank table), random fills of N-UMIs without a target come from one rand() sequence
    in global (cell id, gene) order."""
    s = SynthStream(n_reads=150_000 * SCALE, n_cells=30 * SCALE, n_genes=400, umi_len=6, reads_per_molecule=3)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    side = ()
    if n_rate:
        umi, side = inject_n(umi, gene, n_rate, 13, 6)
        assert len(side) > 50
    kw = dict(cfg_kwargs({"min_before": 5, "min_after": 10}), umi_merge_kind=capi.UMI_MERGE_DIRECTIONAL, max_umi_merge_edit_distance=1,
              umi_merge_multiplier=2.0)
    got = run_group(world, (cb, umi, gene, aux), kw, side)
    c = single((cb, umi, gene, aux), kw, side)
    check(got, c)
    plain = single((cb, umi, gene, aux), cfg_kwargs({"min_before": 5, "min_after": 10}), side)
    assert int(plain.count_matrix_csc(filtered=True)[2].sum()) > int(c.count_matrix_csc(filtered=True)[2].sum())   # -u really merged UMIs
    # (the single context itself is pinned on the oracle for -u: test_gpu_parity.py::test_directional_*)


@pytest.mark.parametrize("n_rate", [0.0, 3e-2])
def test_directional_with_more_shards_than_barcodes(n_rate):
    """-u over 8 shards when only 3 barcodes exist: most shards own no read at all, yet take part in the all-gather of the UMI
    first-occurrence table -- with the key fields the shards agreed on, not a stale or empty layout."""
    s = SynthStream(n_reads=20_000, n_cells=3, n_genes=60, umi_len=6, reads_per_molecule=3, permille_neighbour=0, permille_ambient=0)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    assert len(np.unique(cb)) <= 3
    side = ()
    if n_rate:
        umi, side = inject_n(umi, gene, n_rate, 5, 6)
    kw = dict(cfg_kwargs({"min_before": 2, "min_after": 3}), umi_merge_kind=capi.UMI_MERGE_DIRECTIONAL, max_umi_merge_edit_distance=1,
              umi_merge_multiplier=2.0)
    got = run_group(8, (cb, umi, gene, aux), kw, side)
    c = single((cb, umi, gene, aux), kw, side)
    check(got, c)
    assert len(c.filtered_cells()) >= 2


def test_barcodes_of_several_lengths_and_max_cells():
    """compare_cells orders barcode STRINGS (CellsDataContainer.cpp:329-344): with barcodes of several lengths the packed codes
    do not order like the strings, ties on the sizes must still come out as in one container; -C keeps the largest cells."""
    import test_gpu_stress as ts
    rng = np.random.default_rng(77)
    cb, umi, gene, aux, side = ts.random_stream(rng, n=60_000, n_cb=400, n_gene=6, n_umi=3, cb_len=(8, 11), umi_len=(6, 6), p_nogene=0.05)
    for max_cells in (-1, 25):
        kw = cfg_kwargs({"min_before": 1, "min_after": 2, "max_cells": max_cells})
        got = run_group(2, (cb, umi, gene, aux), kw, side, steps=1)
        c = single((cb, umi, gene, aux), kw, side)
        check(got, c)
        rows = c.cell_rows()
        f = c.filtered_cells().astype(np.int64)
        sizes = list(zip(rows["requested_genes"][f].tolist(), rows["requested_umis"][f].tolist(), rows["total_umis"][f].tolist()))
        assert len(set(sizes)) < len(sizes)                   # the order really depended on the barcode strings
        assert len({len(capi.unpack_code(b, side)) for b in rows["barcode"][f]}) > 1


def test_device_ordering_of_the_global_table(monkeypatch):
    monkeypatch.setenv("DROPEST_DEVICE_SORT_MIN", "1")
    kw, wl, kind, cfg = MERGE_CASES["10x"]
    arrays = parity.canonical_stream(*SynthStream(**kw).generate_host())
    ckw = cfg_kwargs(dict(cfg, merge={"barcodes_kind": kind, "barcodes_file": os.path.join(DATA, wl)}))
    check(run_group(2, arrays, ckw, steps=1), single(arrays, ckw))


def test_one_shard_over_rccl_and_forced_exchange():
    """A world of one process: the RCCL communicator, the grouped send / recv all-to-all (to itself), the staged all-gathers
    and the POSIX shared-memory result buffer are the production code paths of bench.py --gpus N."""
    s = SynthStream(**STREAM)
    run = ShardedRun(s, 0, 1, 0, STREAM["n_reads"], CFG)
    run.shard.set_option("force_exchange", 1)
    for _ in range(2):
        cm, raw, cols = run.step()
    arrays = parity.canonical_stream(*s.generate_host())
    # the device generator numbers genes / chromosomes as the host generator does before canonical_stream: compare with a
    # context fed from the same device arrays
    c = capi.Context(**cfg_kwargs(CFG))
    dev = s.generate_device(0)
    c.push_reads_device(*dev.ptrs, dev.n, adopt=True)
    c.set_initialized(); c.merge_and_filter()
    from dropest_amd.multi import ShardBytes
    # the default: the step ends with the 32-bit slots, widened from the byte form by the shard's host threads under the copy
    assert len(cm) == 4 and run.shard.matrix_form(True) == 3 and run.shard.matrix_form(False) == 3
    slots = {"cm": [x.copy() for x in cm], "raw": [x.copy() for x in raw], "merged": run.merge_pairs}
    check(slots, c)
    assert run.shard.matrix_bytes(False).n_row_listed > 0    # (the byte form stands beside the slots)
    run.shard.set_option("slots_matrix", 0)                  # the step that ends at the byte form
    cm, raw, cols = run.step()
    assert isinstance(cm, ShardBytes) and cm.nnz > 1000 and raw.n_row_listed > 0
    got = {"cm": widen_shard_matrix(cm), "raw": widen_shard_matrix(raw), "merged": run.merge_pairs}
    check(got, c)
    wide = run.shard.matrix(True)                            # ... and the 32-bit accessor widens it on the host
    assert all(np.array_equal(a, b) for a, b in zip(wide, got["cm"]))
    run.shard.set_option("byte_matrix", 0)                   # the 16-bit form (gene ids below 65536) of the same pass
    cm16, raw16, _ = run.step()
    assert len(cm16) == 6 and cm16[1].dtype == np.uint16
    for a, b in ((cm16, cm), (raw16, raw)):
        assert all(np.array_equal(x, y) for x, y in zip(widen_shard_matrix(a), widen_shard_matrix(b)))
    run.shard.set_option("narrow_matrix", 0)                 # ... and the 32-bit one
    cm32, raw32, _ = run.step()
    assert len(cm32) == 4 and all(np.array_equal(x, y) for x, y in zip(raw32, widen_shard_matrix(raw)))
    run.shard.set_option("byte_matrix", 1)
    ph = run.shard.phase_stats()
    assert ph["exchange_record_bytes"]["bytes"] == 12        # barcode + UMI in 64 bits, gene + mark + chromosome in 32
    assert ph["partition"]["steps"] == 5 and ph["all_to_all"]["steps"] == 5
    dev.free()


def test_matrix_forms_of_a_group_agree():
    """Three shards, the three forms the step can write the global matrices in (shard options byte_matrix / narrow_matrix): the
    byte form -- every shard's deltas and values at their global places, its listed entries in its own segment of the shared
    buffer -- decodes to the 16-bit and the 32-bit one; when a shard would have to list more than its segment holds, the matrix is placed
    again in a wider form."""
    s = SynthStream(n_reads=400_000, n_cells=60, n_genes=9000, umi_len=8, permille_neighbour=50)
    arrays = parity.canonical_stream(*s.generate_host())
    arrays[0][::97] = arrays[0][0]; arrays[2][::97] = arrays[2][0]; arrays[3][::97] = arrays[3][0]   # ~4000 UMIs of one gene in one cell: a value beyond a byte
    n = len(arrays[0])
    bounds = [0, n // 5, n // 2, n]
    g = ShardGroup([0, 0, 0], **cfg_kwargs({"min_before": 1, "min_after": 5}))
    for i, sh in enumerate(g.shards):
        sh.set_reads(capi.DeviceArrays.from_host(0, *[a[bounds[i]:bounds[i + 1]] for a in arrays]), bounds[i])
    forms = {}
    for name, opts in (("bytes", {"byte_matrix": 1}), ("u16", {"byte_matrix": 0, "narrow_matrix": 1}), ("u32", {"byte_matrix": 0, "narrow_matrix": 0})):
        for sh in g.shards:
            for k, v in opts.items():
                sh.set_option(k, v)
        g.step()
        s0 = g.shards[0]
        if name == "bytes":
            for filtered in (True, False):
                b = s0.matrix_bytes(filtered)
                assert b.nnz > 5000 and b.n_row_listed > 100 and b.n_value_listed >= 1
            with pytest.raises(capi.DropestError):
                s0.matrix_narrow(True)
        else:
            with pytest.raises(capi.DropestError):
                s0.matrix_bytes(True)
        forms[name] = [[x.copy() for x in s0.matrix(f)] for f in (True, False)]
    for name in ("u16", "u32"):
        for a, b in zip(forms["bytes"], forms[name]):
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), name
    c = single(arrays, cfg_kwargs({"min_before": 1, "min_after": 5}))
    check({"cm": forms["bytes"][0], "raw": forms["bytes"][1], "merged": (np.zeros(0, np.uint64),) * 2}, c)
    assert forms["bytes"][0][2].max() > 1000
    for sh in g.shards:                               # cm_raw planned on the host from the table of all real cells (the fallback)
        sh.set_option("byte_matrix", 1); sh.set_option("raw_on_device", 0)
    g.step()
    ph = g.shards[0].phase_stats()
    assert "order:cm_raw" in ph and ph["raw:plan"]["steps"] == 4
    for f, want in zip((True, False), forms["bytes"]):
        assert all(np.array_equal(x, y) for x, y in zip(g.shards[0].matrix(f), want))
    # a shard that would have to list more than its segment holds: that matrix's columns are placed again in the 16-bit form, by every
    # shard alike (the step must not fail for the shape of the data).  A cap of 16 overflows both matrices, one of 2048 only
    # cm_raw (small real cells list nearly every row); cm_raw planned on the device and on the host.
    # (the step that ENDS at the byte form in the shared buffer: slots_matrix off -- the default slots step has no shared lists, a shard whose
    # own lists overflow places its columns as 32-bit slots by itself: test_gpu_multi_oracle.py)
    for raw_dev in (1, 0):
        for cap, wide in ((16, (True, True)), (2048, (False, True))):
            for sh in g.shards:
                sh.set_option("slots_matrix", 0); sh.set_option("raw_on_device", raw_dev); sh.set_option("byte_list_cap", cap)
            g.step()
            assert "matrix:overflow" in g.shards[0].phase_stats()
            for f, want, w in zip((True, False), forms["bytes"], wide):
                assert all(np.array_equal(x, y) for x, y in zip(g.shards[0].matrix(f), want))
                if w:
                    with pytest.raises(capi.DropestError):
                        g.shards[0].matrix_bytes(f)
                else:
                    assert g.shards[0].matrix_bytes(f).nnz > 5000
    for sh in g.shards:
        sh.set_option("byte_list_cap", 0)
    g.step()                                          # and back: the byte form again
    assert g.shards[0].matrix_bytes(False).n_row_listed > 100 and g.shards[0].matrix_form(False) == 2
    for f, want in zip((True, False), forms["bytes"]):
        assert all(np.array_equal(x, y) for x, y in zip(g.shards[0].matrix(f), want))
    g.close()


@pytest.mark.parametrize("world", [2, 3])
def test_batches_pushed_from_host(world):
    """What the C++ facade does when it owns N GPUs: the stream arrives in batches from HOST memory (dropest_shard_push_reads);
    a shard takes batches up to its quota, then the next shard starts -- ranges of unequal length, the last shard may get
    nothing."""
    s = SynthStream(n_reads=250_000, n_cells=40, n_genes=1500, umi_len=8)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    umi, side = inject_n(umi, gene, 5e-3, 3, 8)
    kw = cfg_kwargs({"min_before": 10, "min_after": 20})
    B = 40_000
    g = ShardGroup([0] * world, **kw)
    for sh in g.shards:
        sh.set_side_strings(side)
    quota = [3, 2][:world - 1]                       # batches per shard; the last shard takes the rest (world 3: two of them, one short)
    shard_of = [i for i, q in enumerate(quota) for _ in range(q)]
    for k, a in enumerate(range(0, len(cb), B)):
        g.shards[shard_of[k] if k < len(shard_of) else world - 1].push_reads(cb[a:a + B], umi[a:a + B], gene[a:a + B], aux[a:a + B], a)
    with pytest.raises(capi.DropestError):           # a batch that does not continue the shard's range
        g.shards[0].push_reads(cb[:10], umi[:10], gene[:10], aux[:10], 5)
    g.step()
    s0 = g.shards[0]
    got = {"cm": [x.copy() for x in s0.matrix(True)], "raw": [x.copy() for x in s0.matrix(False)], "merged": s0.merged_barcodes()}
    g.close()
    check(got, single((cb, umi, gene, aux), kw, side))


def test_push_reads_from_pinned_and_pageable_memory_agree():
    """dropest_push_reads copies from pinned arrays in place and stages pageable ones: same container either way, also when
    the device arrays have to grow between batches."""
    import ctypes as C
    s = SynthStream(n_reads=3_000_000, n_cells=50, n_genes=2000)
    arrays = [np.ascontiguousarray(a) for a in parity.canonical_stream(*s.generate_host())]
    kw = cfg_kwargs(CFG)
    ref = single(arrays, kw)                                     # pageable numpy memory: the staged path
    L = capi.lib()
    for a in arrays:                                             # page-lock the same arrays (hipHostRegister): the in-place path
        d = C.c_void_p()
        assert L.dropest_host_register(0, a.ctypes.data, a.nbytes, C.byref(d)) == 0
    try:
        c = capi.Context(**kw)
        for at in range(0, len(arrays[0]), 700_000):
            c.push_reads(*[a[at:at + 700_000] for a in arrays])
        c.set_initialized(); c.merge_and_filter()
        for filt in (True, False):
            for x, y in zip(ref.count_matrix_csc(filtered=filt), c.count_matrix_csc(filtered=filt)):
                assert np.array_equal(x, y)
    finally:
        for a in arrays:
            L.dropest_host_unregister(0, a.ctypes.data)


@pytest.mark.parametrize("n_parts", [1, 2, 3, 8, 64, 256])
def test_partition_by_owner_is_stable_and_complete(n_parts):
    """dropest_partition_by_owner: reads grouped by owner(cb) = mix64(cb) mod n, each owner's reads in stream order, every
    read exactly once with its position -- at a size that is not a multiple of the tile."""
    import ctypes as C
    s = SynthStream(n_reads=1_000_003, n_cells=300, n_genes=2000)
    arrays = s.generate_host()
    n = len(arrays[0])
    L = capi.lib()
    src = capi.DeviceArrays.from_host(0, *arrays)
    dst = capi.DeviceArrays(0, n)
    idx = C.c_void_p(); scratch = C.c_void_p()
    need = C.c_uint64()
    assert L.dropest_partition_scratch_bytes(n, C.byref(need)) == 0
    assert L.dropest_dev_alloc(0, n * 4, C.byref(idx)) == 0 and L.dropest_dev_alloc(0, max(need.value, 1), C.byref(scratch)) == 0
    counts = np.zeros(n_parts, np.uint64)
    rc = L.dropest_partition_by_owner(0, *src.ptrs, n, n_parts, *dst.ptrs, idx, counts.ctypes.data, scratch, need.value)
    assert rc == 0, L.dropest_last_error()
    out = dst.to_host()
    pos = np.zeros(n, np.uint32)
    assert L.dropest_dev_copy_to_host(0, pos.ctypes.data, idx, n * 4) == 0
    owner = np.array([L.dropest_owner_of(int(c), n_parts) for c in arrays[0][:2000]])      # the library's own owner function ...
    import test_multi_gloo as tg
    all_owner = (tg.mix64(arrays[0].copy()) % np.uint64(n_parts)).astype(np.int64)           # ... equals mix64 mod n
    assert np.array_equal(owner, all_owner[:2000])
    assert np.array_equal(counts.astype(np.int64), np.bincount(all_owner, minlength=n_parts))
    want = np.argsort(all_owner, kind="stable")
    assert np.array_equal(pos.astype(np.int64), want)
    for a, b in zip(arrays, out):
        assert np.array_equal(a[want], b)
    for p in (idx, scratch):
        L.dropest_dev_free(0, p)
    src.free(); dst.free()


def _quality_table(ctx, side=(), lengths=False):
    """{(barcode, gene, UMI): (reads, quality sums[, quality length])} of every real cell of a context."""
    rows = ctx.cell_rows()
    out = {}
    for cell in np.flatnonzero(rows["is_real"].astype(bool) & ~rows["is_merged"].astype(bool)):
        g, u, r, m = ctx.cell_molecules(int(cell))
        q = ctx.cell_molecule_qualities(int(cell), len(g))
        ql = ctx.cell_molecule_quality_lengths(int(cell), len(g)) if lengths else None
        for j in range(len(g)):
            out[(int(rows["barcode"][cell]), int(g[j]), capi.unpack_code(u[j], side))] = (int(r[j]), tuple(int(x) for x in q[j])) + ((int(ql[j]),) if lengths else ())
    return out


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("seed", range(6))
def test_quality_strings_of_several_lengths_across_shards(seed, world):
    """dropest_shard_set_umi_qualities_var: a quality length per molecule (UMI.cpp:21-34), the lengths travel with the reads (one byte each)
    and with the molecule rows of a barcode merge; through the simple CB merge, N-UMIs and -u.  Sums and lengths per molecule as one context."""
    import ctypes
    from test_gpu_stress import random_stream
    from test_gpu_quality import _molecule_lengths, _qualities
    rng = np.random.default_rng(13000 + seed)
    cb, umi, gene, aux, side = random_stream(
        rng, n=int(rng.integers(500, 6000)), n_cb=int(rng.integers(2, 40)), n_gene=int(rng.integers(1, 12)),
        n_umi=int(rng.integers(2, 60)), cb_len=(8, 8), umi_len=(6, 6), n_rate=0.05 if seed % 2 else 0.0)
    stride = 9 if seed % 2 else 6
    lens = _molecule_lengths(cb, umi, gene, seed, 0 if seed == 3 else 2, stride)
    qual = _qualities(len(cb), stride, 200 + seed)
    directional = seed % 3 == 0
    kw = dict(merge_kind=capi.MERGE_SIMPLE, max_cb_merge_edit_distance=int(rng.integers(1, 6)), min_merge_fraction=float(rng.choice([0.0, 0.1, 0.3])),
              min_genes_before_merge=int(rng.integers(0, 3)), min_genes_after_merge=0,
              umi_merge_kind=capi.UMI_MERGE_DIRECTIONAL if directional else capi.UMI_MERGE_SIMPLE)
    libc = ctypes.CDLL("libc.so.6")
    c = capi.Context(**kw)
    if side:
        c.set_side_strings(side)
    c.push_reads(cb, umi, gene, aux)
    c.set_umi_qualities(qual, lens)
    c.set_initialized()
    libc.srand(1)
    c.merge_and_filter()
    want = _quality_table(c, side, lengths=True)
    n = len(cb)
    bounds = [n * i // world for i in range(world + 1)]
    g = ShardGroup([0] * world, **kw)
    for i, sh in enumerate(g.shards):
        if side:
            sh.set_side_strings(side)
        lo, hi = bounds[i], bounds[i + 1]
        sh.set_reads(capi.DeviceArrays.from_host(0, cb[lo:hi], umi[lo:hi], gene[lo:hi], aux[lo:hi]), lo)
        sh.set_umi_qualities(qual[lo:hi], lens[lo:hi])
    for _ in range(2):
        libc.srand(1)
        g.step()
    got = {}
    for sh in g.shards:
        got.update(_quality_table(sh.ctx, side, lengths=True))
    assert len(got) == len(want) and len({v[2] for v in want.values()}) > 1
    bad = [k for k in want if got.get(k) != want[k]]
    assert not bad, (len(bad), bad[0], got.get(bad[0]), want[bad[0]])
    check({"cm": [x.copy() for x in g.shards[0].matrix(True)], "raw": [x.copy() for x in g.shards[0].matrix(False)], "merged": g.shards[0].merged_barcodes()}, c)
    g.close()


def test_a_wrong_quality_length_on_a_shard_is_the_reference_exception():
    """A read whose string has another length than its molecule's (UMI.cpp:26-28), on the second of two shards: the step fails with the text."""
    P = capi.pack_seq
    cbs = [P("ACGTACGTACGT"), P("TTGTACGTACGA")]
    cb = np.array([cbs[0], cbs[1], cbs[0], cbs[1], cbs[0], cbs[1]], np.uint64)
    umi = np.array([P("ACGTAC")] * 6, np.uint64)
    gene = np.zeros(6, np.uint32); aux = np.full(6, 2 << 16, np.uint32)
    qual = np.full((6, 8), 40, np.uint8)
    lens = np.array([5, 8, 5, 8, 6, 8], np.uint8)
    g = ShardGroup([0, 0])
    for i, (lo, hi) in enumerate(((0, 3), (3, 6))):
        g.shards[i].set_reads(capi.DeviceArrays.from_host(0, cb[lo:hi], umi[lo:hi], gene[lo:hi], aux[lo:hi]), lo)
        g.shards[i].set_umi_qualities(qual[lo:hi], lens[lo:hi])
    with pytest.raises(capi.DropestError, match="Wrong quality length: 6, expected: 5"):
        g.step()
    g.close()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("variant", ["plain", "n_umis", "directional"])
def test_umi_qualities_travel_with_the_reads(world, variant):
    """UMI quality strings in a sharded run: every shard is given the strings of ITS reads, they follow the reads through the exchange,
    the sums are accumulated where the cell lives and follow the UMI merges there -- same sums per molecule as one context."""
    import ctypes
    s = SynthStream(n_reads=200_000 * SCALE, n_cells=40 * SCALE, n_genes=800, umi_len=8)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    side = ()
    if variant != "plain":
        umi, side = inject_n(umi, gene, 5e-3, 11, 8)
    qual = np.random.default_rng(77).integers(33, 75, size=(len(cb), 8), dtype=np.uint8)
    kw = cfg_kwargs({"min_before": 10, "min_after": 20})
    if variant == "directional":
        kw["umi_merge_kind"] = capi.UMI_MERGE_DIRECTIONAL
    libc = ctypes.CDLL("libc.so.6")
    c = capi.Context(**kw)
    if side:
        c.set_side_strings(side)
    c.push_reads(cb, umi, gene, aux)
    c.set_umi_qualities(qual)
    c.set_initialized()
    libc.srand(1)
    c.merge_and_filter()
    want = _quality_table(c, side)
    n = len(cb)
    bounds = [n * i // world for i in range(world + 1)]
    g = ShardGroup([0] * world, **kw)
    for i, sh in enumerate(g.shards):
        if side:
            sh.set_side_strings(side)
        sh.set_reads(capi.DeviceArrays.from_host(0, cb[bounds[i]:bounds[i + 1]], umi[bounds[i]:bounds[i + 1]], gene[bounds[i]:bounds[i + 1]], aux[bounds[i]:bounds[i + 1]]), bounds[i])
        sh.set_umi_qualities(qual[bounds[i]:bounds[i + 1]])
    for _ in range(2):
        libc.srand(1)
        g.step()
    got = {}
    for sh in g.shards:
        got.update(_quality_table(sh.ctx, side))
    assert len(want) > 5000 and len(got) == len(want)
    bad = [k for k in want if got.get(k) != want[k]]
    assert not bad, (len(bad), bad[0], got.get(bad[0]), want[bad[0]])
    check({"cm": [x.copy() for x in g.shards[0].matrix(True)], "raw": [x.copy() for x in g.shards[0].matrix(False)], "merged": g.shards[0].merged_barcodes()}, c)
    g.close()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("poisson", [False, True, "simple", "all"])
def test_umi_qualities_follow_a_barcode_merge_across_shards(world, poisson):
    """-m / -M with a whitelist and UMI qualities over 2 / 3 shards: the sums rows of the molecules that change shards travel with them; a
    molecule the target already has keeps the target's sums, one that only merged cells had takes those of the first of them in merge
    order (Gene::merge, Gene.cpp:26-36) -- whichever shards those cells lived on.  Same sums per molecule as one context."""
    kw_s, wl, kind, cfg = MERGE_CASES["10x"]
    s = SynthStream(**dict(kw_s, n_genes=60, umi_len=6, permille_neighbour=250))     # few genes and UMIs: many molecules shared between merged cells
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    qual = np.random.default_rng(78).integers(33, 75, size=(len(cb), 6), dtype=np.uint8)
    ckw = cfg_kwargs(dict(cfg, merge={"barcodes_kind": kind, "barcodes_file": os.path.join(DATA, wl)}))
    if poisson is True:
        ckw.update(merge_kind=capi.MERGE_POISSON_REAL)
    elif poisson == "simple":   # (the whitelist-free merges share the travel of the sums rows: shard_merge_free.h)
        ckw = dict(merge_kind=capi.MERGE_SIMPLE, max_cb_merge_edit_distance=2, min_merge_fraction=0.1, min_genes_before_merge=3, min_genes_after_merge=20)
    elif poisson == "all":
        ckw = dict(merge_kind=capi.MERGE_ALL, max_cb_merge_edit_distance=2, min_genes_before_merge=3, min_genes_after_merge=20)
    c = capi.Context(**ckw)
    c.push_reads(cb, umi, gene, aux)
    c.set_umi_qualities(qual)
    c.set_initialized(); c.merge_and_filter()
    want = _quality_table(c)
    n = len(cb)
    bounds = [n * i // world for i in range(world + 1)]
    g = ShardGroup([0] * world, **ckw)
    for i, sh in enumerate(g.shards):
        sh.set_reads(capi.DeviceArrays.from_host(0, cb[bounds[i]:bounds[i + 1]], umi[bounds[i]:bounds[i + 1]], gene[bounds[i]:bounds[i + 1]], aux[bounds[i]:bounds[i + 1]]), bounds[i])
        sh.set_umi_qualities(qual[bounds[i]:bounds[i + 1]])
    for _ in range(2):
        g.step()
    got = {}
    for sh in g.shards:
        got.update(_quality_table(sh.ctx))
    merged = check({"cm": [x.copy() for x in g.shards[0].matrix(True)], "raw": [x.copy() for x in g.shards[0].matrix(False)], "merged": g.shards[0].merged_barcodes()}, c)
    owner = lambda b: capi.lib().dropest_owner_of(int(b), world)           # noqa: E731
    assert sum(owner(a) != owner(b) for a, b in merged.items()) > 5        # molecules really changed shards
    assert len(want) > 2000 and len(got) == len(want)
    bad = [k for k in want if got.get(k) != want[k]]
    assert not bad, (len(bad), bad[0], got.get(bad[0]), want[bad[0]])
    # and not trivially: some kept molecule's sums differ from what its reads alone in the target would give (a fold happened)
    g.close()


@pytest.mark.parametrize("n,cells", [(125_000_000, 5000), (500_000_000, 20000)], ids=["per_gpu_share", "whole_workload"])
def test_c4_single_context_and_four_shards(n, cells):
    """BASELINE configs[3] ("C4": inDrop v3, split 8 + 8 bp barcode, 8 bp UMI, -m + the inDrop v3 whitelist; configs/indrop_v3.xml:22-29) at
    the size ONE of its four GPUs holds (1.25e8 reads, 5 000 cells) and as the WHOLE workload (5e8 reads, 20 000 cells: 12 GB of reads, which
    one MI355X holds).  Far beyond the oracle, so (a) size-independent properties of the single context -- every read counted once, targets
    final whitelist cells, CSC structure -- and (b) the same stream cut into four contiguous ranges on four in-process shards (all on this
    GPU: partition by owner, all-to-all, sharded whitelist merge, device-planned cm_raw, every shard's columns widened into the shared
    slots) must assemble the same two matrices, column barcodes and merged barcodes, entry for entry."""
    world = 4
    s = SynthStream(n_reads=n, n_cells=cells, n_genes=30000, cb_len=16, whitelist="indrop_v3", umi_len=8, stream_id=4)
    cfg = {"min_before": 20, "min_after": 100, "merge": {"barcodes_kind": capi.BARCODES_CONST, "barcodes_file": os.path.join(DATA, "indrop_v3")}}
    kw = cfg_kwargs(cfg)
    dev = s.generate_device(0)
    c = capi.Context(**kw)
    c.push_reads_device(*dev.ptrs, dev.n, adopt=True)
    c.set_initialized()
    n_real_before = c.real_cells_number()
    c.merge_and_filter()
    rows = c.cell_rows()
    merged = rows["is_merged"].astype(bool)
    assert int(rows["total_reads"].astype(np.int64)[~merged].sum()) + int(c.global_counters()[0]) == n      # every read once
    mt = c.merge_targets().astype(np.int64)
    src = np.flatnonzero(mt != np.arange(len(mt)))
    assert len(src) > 1000 and np.all(mt[mt[src]] == mt[src]) and c.real_cells_number() < n_real_before
    wl_cells = set(int(x) for x in s.cell_cb)
    assert all(int(b) in wl_cells for b in rows["barcode"][mt[src[:20000]]])
    want = {}
    for filt, name in ((True, "cm"), (False, "raw")):
        p, i, x = c.count_matrix_csc(filtered=filt)
        d = np.diff(i.astype(np.int64)); d[p[1:-1].astype(np.int64) - 1] = 1
        assert np.all(d > 0) and np.all(x > 0) and len(i) > 1_000_000                                        # genes ascend inside a column
        want[name] = (p.astype(np.uint64).copy(), i.copy(), x.copy())
    want["cm_bc"] = [int(rows["barcode"][int(k)]) for k in c.filtered_cells()]
    want["raw_bc"] = [int(b) for b in rows["barcode"][rows["is_real"].astype(bool)]]
    want["merged"] = {int(rows["barcode"][k]): int(rows["barcode"][int(mt[k])]) for k in src}
    c.close(); dev.free()
    # four shards on this GPU, each with its contiguous quarter of the stream
    g = ShardGroup([0] * world, **kw)
    bounds = [n * k // world for k in range(world + 1)]
    for k, sh in enumerate(g.shards):
        sh.set_reads(s.generate_device(0, first=bounds[k], n=bounds[k + 1] - bounds[k]), bounds[k])
    g.step()
    s0 = g.shards[0]
    for filt, name in ((True, "cm"), (False, "raw")):
        gp, gi, gx, gb = s0.matrix(filt)
        assert np.array_equal(gp.astype(np.uint64), want[name][0]) and np.array_equal(gi, want[name][1]) and np.array_equal(gx, want[name][2]), name
        assert [int(b) for b in gb] == want[name + "_bc"], name
    ms, mt2 = s0.merged_barcodes()
    assert dict(zip((int(b) for b in ms), (int(b) for b in mt2))) == want["merged"]
    assert s0.phase_stats()["all_to_all"]["bytes"] > 0
    g.close()
