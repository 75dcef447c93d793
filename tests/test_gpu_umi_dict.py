"""Molecule keys of two words (missing 5): the reference indexes genes and UMIs with size_t (StringIndexer.cpp:10-18), one
context here sorts cell | gene | UMI in ONE 64-bit word.  When the gene and UMI fields alone reach 64 bits the pass builds a dictionary of the
stream's UMIs on the device and the key carries a UMI's rank in it (csrc/k_umidict.h, dropest_set_umi_dictionary).  Two kinds of cases:
streams that NEED the dictionary (26..30-base UMIs beside up to 2^20 gene ids) against the oracle, and the suite's own merge / UMI-merge /
quality / mutator cases run again with the dictionary forced (DROPEST_UMI_DICT=2): every consumer of the key's UMI field must cope with ranks."""
import os

import numpy as np
import pytest

from dropest_amd import capi
from dropest_amd.multi import ShardGroup
from dropest_amd.synth import SynthStream, inject_n
from oracle import Oracle

import parity
import test_gpu_parity as tp
import test_gpu_stress as ts

pytestmark = pytest.mark.gpu
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dropest_amd", "data", "barcodes")


def used_dictionary(c):
    return c.kernel_stats().get("count:umi_dictionary", {"launches": 0})["launches"] >= 1


def wide_stream(seed, n=40_000, n_rate=0.0, umi_len=(30, 30), n_gene=3000, spread_genes=True):
    """UMIs of 26..30 bases; gene ids pushed up to 2^20 by unused ids in between would not survive canonical_stream (first-seen dense), so the
    width comes from the UMI (60 bits + sentinel) and a few thousand genes (12 bits): gene + UMI >= 64 already without a cell bit."""
    rng = np.random.default_rng(seed)
    return ts.random_stream(rng, n=n, n_cb=400, n_gene=n_gene, n_umi=2500, cb_len=(16, 16), umi_len=umi_len, n_rate=n_rate, p_nogene=0.05)


@pytest.mark.parametrize("n_rate,umi_len", [(0.0, (30, 30)), (0.02, (30, 30)), (0.0, (26, 31)), (0.0, (31, 31))])
def test_gene_and_umi_fields_of_64_bits_and_more(n_rate, umi_len):
    cb, umi, gene, aux, side = wide_stream(7000 + int(n_rate * 1000) + umi_len[0], n_rate=n_rate, umi_len=umi_len)
    o = parity.oracle_run(Oracle, dict(min_genes_before=2, min_genes_after=4), cb, umi, gene, aux, side)
    c = parity.gpu_run(dict(min_genes_before_merge=2, min_genes_after_merge=4), cb, umi, gene, aux, side, chunks=2, profile=True)
    cell, g, u = c.key_width()
    assert used_dictionary(c) and u <= 13 and cell + g + u <= 64          # ~2 500 distinct UMIs (+ escapes): 12-13 bits instead of 61
    parity.compare(o, c, side)
    assert len(c.filtered_cells()) > 50


def test_wide_fields_with_the_sampled_key_plan(monkeypatch):
    """The large-stream path: layout planned from every 256th read, exact statistics gathered by the key pass (which then reads RANKS)."""
    monkeypatch.setenv("DROPEST_CB_SAMPLE_MIN", "1000")
    cb, umi, gene, aux, side = wide_stream(7100, n=50_000, n_rate=0.01)
    o = parity.oracle_run(Oracle, dict(min_genes_before=2, min_genes_after=4), cb, umi, gene, aux, side)
    c = parity.gpu_run(dict(min_genes_before_merge=2, min_genes_after_merge=4), cb, umi, gene, aux, side, profile=True)
    assert used_dictionary(c)
    parity.compare(o, c, side)


def test_the_one_long_umi_the_sample_did_not_see(monkeypatch):
    """A stream of 8-base UMIs whose only 31-base UMI sits where neither sample looks: the plan from the sample fits 64 bits, the exact one
    does not -- the dictionary is built behind the first key pass and the keys are made again."""
    monkeypatch.setenv("DROPEST_CB_SAMPLE_MIN", "1000")
    rng = np.random.default_rng(7200)
    cb, umi, gene, aux, side = ts.random_stream(rng, n=30_000, n_cb=300, n_gene=3000, n_umi=400, cb_len=(12, 12), umi_len=(8, 8), p_nogene=0.05)
    at = 12_345
    while gene[at] == capi.NO_GENE or at % 256 == 0:
        at += 1
    umi[at] = capi.pack_seq("ACGTTGCAAGCTTCGATTGACCATGCATGCA")         # 31 bases: 62 bits + sentinel
    cb, umi, gene, aux = parity.canonical_stream(cb, umi, gene, aux)
    o = parity.oracle_run(Oracle, dict(min_genes_before=1, min_genes_after=2), cb, umi, gene, aux, side)
    c = parity.gpu_run(dict(min_genes_before_merge=1, min_genes_after_merge=2), cb, umi, gene, aux, side, profile=True)
    st = c.kernel_stats()
    assert used_dictionary(c) and st.get("count:key_plan_redone", {"launches": 0})["launches"] >= 1
    parity.compare(o, c, side)


def test_wide_fields_whitelist_merge_n_umis_and_a_second_pass():
    s = SynthStream(n_reads=200_000, n_cells=40, n_genes=2000, umi_len=30, permille_neighbour=150)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    umi, side = inject_n(umi, gene, 2e-3, 5, 30)
    wl = os.path.join(DATA, "10x_aug_2016_split")
    o = parity.oracle_run(Oracle, dict(merge_kind=1, barcodes_kind=capi.BARCODES_CONST, barcodes_file=wl, min_genes_before=3, min_genes_after=20),
                          cb, umi, gene, aux, side)
    c = parity.gpu_run(dict(merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=capi.BARCODES_CONST, barcodes_file=wl, min_genes_before_merge=3,
                            min_genes_after_merge=20), cb, umi, gene, aux, side, profile=True)
    assert used_dictionary(c)
    parity.compare(o, c, side)
    assert int((c.merge_targets() != np.arange(c.total_cells_number())).sum()) > 20
    c.reset_results()                                                     # the same reads once more: the dictionary is rebuilt
    c.set_initialized(); c.merge_and_filter()
    parity.compare(o, c, side)


def test_wide_fields_directional_umi_merge():
    """-u on 28-base UMIs drawn near each other: ranks say nothing about bases, every group is decided by the host's replay."""
    rng = np.random.default_rng(7300)
    bases = list("ACGT")
    roots = ["".join(rng.choice(bases, 28)) for _ in range(40)]
    umis = []
    for r in roots:
        umis.append(r)
        for _ in range(4):
            i = int(rng.integers(0, 28))
            umis.append(r[:i] + str(rng.choice(bases)) + r[i + 1:])
    cbs = ["".join(rng.choice(bases, 12)) for _ in range(40)]
    n = 30_000
    cb = np.array([capi.pack_seq(cbs[k]) for k in rng.integers(0, 40, n)], np.uint64)
    w = 1.0 / np.arange(1, len(umis) + 1) ** 0.7
    umi = np.array([capi.pack_seq(umis[i]) for i in rng.choice(len(umis), n, p=w / w.sum())], np.uint64)
    gene = np.where(rng.random(n) < 0.5, rng.integers(0, 10, n), rng.integers(0, 1000, n)).astype(np.uint32)   # a few hot genes: large groups
    aux = (rng.integers(0, 5, n) | (2 << 16)).astype(np.uint32)
    cb, umi, gene, aux = parity.canonical_stream(cb, umi, gene, aux)
    o, c = tp._both_directional(cb, umi, gene, aux, mult=1.5, max_ed=1, min_genes=2)
    cell, g, u = c.key_width()
    assert used_dictionary(c) and u <= 9 and g >= 10              # 56 + 10 bits without the dictionary
    has_gene = gene != capi.NO_GENE
    distinct = np.unique(np.stack([cb[has_gene], gene[has_gene].astype(np.uint64), umi[has_gene]]), axis=1).shape[1]
    assert int(c.molecules()[0].shape[0]) < distinct * 0.98


def test_wide_fields_in_a_split_run_share_one_dictionary():
    """Round 6: the shards of a split / sharded run gather ONE dictionary (shard_run.h: global_umi_dictionary): ranks mean the same on every
    shard.  The split of a context against one context over the same reads (which other tests pin on the oracle)."""
    cb, umi, gene, aux, side = wide_stream(7400, n=4000)
    kw = dict(min_genes_before_merge=0, min_genes_after_merge=0)
    one = parity.gpu_run(kw, cb, umi, gene, aux, side)
    c = capi.Context(**kw)
    if side:
        c.set_side_strings(side)
    c.push_reads(cb, umi, gene, aux)
    g = ShardGroup.split(c, 2)
    g.step()
    s0 = g.shards[0]
    assert s0.phase_stats().get("umi_dictionary", {"steps": 0})["steps"] == 1
    for filt in (True, False):
        p, i, x, b = s0.matrix(filt)
        p1, i1, x1 = one.count_matrix_csc(filtered=filt)
        assert np.array_equal(p.astype(np.uint64), p1.astype(np.uint64)) and np.array_equal(i, i1) and np.array_equal(x, x1)
    g.close()


def test_mode_1_one_context_instead_of_a_split():
    """cell + gene + UMI = 11 + 15 + 41 bits: refused by default ("sort key needs", dropest_ctx_split), one context with mode 1."""
    rng = np.random.default_rng(4100)
    cb, umi, gene, aux, side = ts.random_stream(rng, n=60_000, n_cb=1500, n_gene=20_000, n_umi=3000, cb_len=(14, 14), umi_len=(20, 20), n_rate=0.02,
                                                p_nogene=0.05)
    c = capi.Context(min_genes_before_merge=2, min_genes_after_merge=4)
    c.set_side_strings(side)
    c.push_reads(cb, umi, gene, aux)
    with pytest.raises(capi.DropestError) as e:
        c.set_initialized()
    assert "sort key needs" in str(e.value)
    c = capi.Context(min_genes_before_merge=2, min_genes_after_merge=4)
    c.set_umi_dictionary(1)
    c.set_profiling(True)
    c.set_side_strings(side)
    c.push_reads(cb, umi, gene, aux)
    c.set_initialized(); c.merge_and_filter()
    assert used_dictionary(c)
    o = parity.oracle_run(Oracle, dict(min_genes_before=2, min_genes_after=4), cb, umi, gene, aux, side)
    parity.compare(o, c, side)


# ---- the suite's own cases with the dictionary forced: every reader of the key's UMI field ----------------------------------------------

@pytest.fixture
def forced(monkeypatch):
    monkeypatch.setenv("DROPEST_UMI_DICT", "2")
    return monkeypatch


@pytest.mark.parametrize("seed", range(4))
def test_forced_random_small_streams(forced, seed):
    ts.test_random_small_streams(seed)


@pytest.mark.parametrize("n", [1, 2, 4096, 4097, 16385])
def test_forced_tile_boundary_sizes(forced, n):
    ts.test_tile_boundary_sizes(n)


def test_forced_shapes_and_edges(forced):
    tp.test_c2_shape_100k()
    tp.test_single_read_and_empty()
    tp.test_intergenic_only_cell_and_ragged()
    tp.test_query_levels_and_reads_output()
    ts.test_one_giant_molecule_and_all_distinct_barcodes()
    ts.test_variable_lengths_and_ns()


@pytest.mark.parametrize("layout", sorted(tp.LAYOUTS))
def test_forced_sort_layouts(forced, layout):
    for case in ("plain", "n_umis", "cb_merge"):
        tp.test_sort_layouts_agree_with_oracle(forced, layout, case)


def test_forced_whitelist_merges_and_n_umis(forced):
    tp.test_reference_fixture_merge_by_real_barcodes()
    tp.test_c3_shape_merge_10x_whitelist()
    tp.test_reference_fixture_umi_merge_strategy_simple()
    tp.test_survey_probe_random_fill()
    tp.test_n_umis_synthetic(1e-2, 120_000)
    tp.test_n_umis_with_cb_merge()


def test_forced_directional(forced):
    tp.test_directional_reference_fixture_on_gpu()
    tp.test_directional_with_n_umis_and_cb_merge()
    for seed in range(3):
        ts.test_random_directional_umi_merge(seed)


def test_forced_merges_without_a_whitelist(forced):
    tp.test_simple_merge_synthetic(2, 0.2, 8)
    tp.test_simple_merge_ties_are_replayed()
    tp.test_simple_merge_with_n_and_directional()
    tp.test_poisson_simple_merge_synthetic(2, 1e-7, 10)
    tp.test_merge_all_synthetic(2)
    for seed in range(2):
        ts.test_random_simple_merge(seed)
        ts.test_random_poisson_simple_merge(seed)


def test_forced_poisson_collisions_and_distributions(forced):
    tp.test_poisson_reference_fixture_on_gpu()
    tp.test_poisson_merge_10x_whitelist()
    tp.test_umi_distribution_matches_oracle()
    tp.test_collisions_adjuster_table()


def test_forced_mutators_and_mark_queries(forced):
    tp.test_public_mutators_exclude_merge_cells_merge_umis()
    tp.test_count_matrices_under_other_mark_queries(True)
    tp.test_prefetched_raw_matrix_is_the_same_matrix()


def test_forced_umi_qualities(forced):
    import test_gpu_quality as tq
    for name in sorted(dir(tq)):
        f = getattr(tq, name)
        if name.startswith("test_") and callable(f) and not getattr(f, "pytestmark", None) and f.__code__.co_argcount == 0:
            f()
