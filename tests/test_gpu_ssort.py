"""The splitter sort (dropest_amd/csrc/k_ssort.h: sampled splitters, two partitions, LDS-resident finishing sort fused with
the reads -> molecules reduce) against the oracle, forced on streams of every size -- by default it only takes streams of
2^22 reads and more -- and against the LSD path on the same resident stream at BASELINE size."""
import os

import numpy as np
import pytest

from dropest_amd import capi
from dropest_amd.synth import SynthStream
from oracle import Oracle

import parity
import test_gpu_parity as tp
import test_gpu_stress as ts

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["fused_keys", "two_kernels"])
def splitter(monkeypatch, request):
    """The splitter sort forced on any size, both ways its first partition can run: fused into the key pass (k_keyscatter.h: the sample
    comes from the reads; by default only when the hot list covers most reads) and as build_keys + ss_scatter_res_l1."""
    monkeypatch.setenv("DROPEST_SORT", "splitter")
    if request.param == "fused_keys":
        monkeypatch.setenv("DROPEST_FUSED_KEYS_MIN_COVERAGE", "0")
    else:
        monkeypatch.setenv("DROPEST_NO_FUSED_KEYS", "1")
    return request.param


def test_the_key_pass_partitions_when_asked(splitter):
    s = SynthStream(n_reads=300_000, n_cells=60, n_genes=3000)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    c = parity.gpu_run(dict(min_genes_before_merge=10, min_genes_after_merge=30), cb, umi, gene, aux, profile=True)
    k = c.kernel_stats()
    assert ("build_keys+L1" in k) == (splitter == "fused_keys") and ("ss_scatter:L1:keys" in k) == (splitter == "two_kernels")
    o = parity.oracle_run(Oracle, dict(min_genes_before=10, min_genes_after=30), cb, umi, gene, aux)
    parity.compare(o, c)


def test_c2_shapes_use_the_splitter_path(splitter):
    o, c = tp._both(dict(n_cells=40, n_genes=3000), 100_000, 20, 100)
    assert c.sort_layout()["sort"] == "splitter" and c.sort_layout()["value_bytes"] == 0
    tp._both(dict(n_cells=200, n_genes=8000), 1_000_000, 20, 100, chunks=7)
    tp._both(dict(n_cells=8, n_genes=50, umi_len=6), 2000, 3, 5)
    tp._both(dict(n_cells=30, n_genes=1500), 60_000, 10, 10, levels="e", reads_output=True)


@pytest.mark.parametrize("tb", [9, 13, 19, 20])
def test_every_fan_out(splitter, monkeypatch, tb):
    """F1 x F2 buckets with F1 = 2^(tb / 2), F2 = 2^(tb - tb / 2): odd splits, and the 1024-bucket kernels (tb 19, 20) that streams
    beyond 4e8 reads take, forced on a small stream (most buckets empty or a handful of records)."""
    monkeypatch.setenv("DROPEST_SSORT_TB", str(tb))
    o, c = tp._both(dict(n_cells=60, n_genes=3000), 300_000, 10, 30)
    assert c.sort_layout()["sort"] == "splitter"
    if tb == 20:
        monkeypatch.setenv("DROPEST_FORCE_BYTE_VALUES", "1")
        tp._both(dict(n_cells=60, n_genes=4000, umi_len=12), 150_000, 20, 50)


def test_key_plus_mark_byte_layout(splitter, monkeypatch):
    """C3's layout (the key uses all 64 bits, the mark travels as one byte) forced on a small v3 stream."""
    monkeypatch.setenv("DROPEST_FORCE_BYTE_VALUES", "1")
    o, c = tp._both(dict(n_cells=60, n_genes=4000, umi_len=12), 150_000, 20, 50)
    assert c.sort_layout()["sort"] == "splitter" and c.sort_layout()["value_bytes"] == 1


def test_rebased_keys_behind_the_first_partition(monkeypatch):
    """Key + mark byte layout: the first partition hands on one word per record -- the key counted from its coarse bucket's lower splitter,
    the mark in the three bits that frees -- so the second partition and the finishing sort run on keys only.  On, off
    (DROPEST_SS_NO_REBASE) and with a difference that does not fit (DROPEST_SS_REBASE_BITS: the counting partitions redo the pass): the
    oracle's results every time."""
    monkeypatch.setenv("DROPEST_SORT", "splitter")
    monkeypatch.setenv("DROPEST_NO_FUSED_KEYS", "1")
    monkeypatch.setenv("DROPEST_FORCE_BYTE_VALUES", "1")
    cb, umi, gene, aux = parity.canonical_stream(*SynthStream(n_reads=400_000, n_cells=80, n_genes=4000, umi_len=12).generate_host())
    o = parity.oracle_run(Oracle, dict(min_genes_before=10, min_genes_after=30), cb, umi, gene, aux)
    for mode in ("rebase", "off", "does_not_fit"):
        if mode == "off":
            monkeypatch.setenv("DROPEST_SS_NO_REBASE", "1")
        if mode == "does_not_fit":
            monkeypatch.delenv("DROPEST_SS_NO_REBASE")
            monkeypatch.setenv("DROPEST_SS_REBASE_BITS", "12")
        c = parity.gpu_run(dict(min_genes_before_merge=10, min_genes_after_merge=30), cb, umi, gene, aux, profile=True)
        names = set(c.kernel_stats())
        assert c.sort_layout()["sort"] == "splitter" and c.sort_layout()["value_bytes"] == 1
        assert "ss_scatter:L1:key+1B" in names
        if mode == "rebase":
            assert "ss_scatter:L2:keys" in names and "ss_local:keys" in names and "count:ss_reserve_overflow" not in names, names
        if mode == "off":
            assert "ss_scatter:L2:key+1B" in names and "ss_local:key+1B" in names, names
        if mode == "does_not_fit":
            assert "count:ss_reserve_overflow" in names and "ss_hist:L1" in names, names
        parity.compare(o, c)


def test_general_layout_keeps_the_lsd_sort(splitter, monkeypatch):
    monkeypatch.setenv("DROPEST_FORCE_GENERAL_LAYOUT", "1")
    o, c = tp._both(dict(n_cells=40, n_genes=3000), 100_000, 20, 100)
    assert c.sort_layout()["value_bytes"] == 4 and c.sort_layout()["sort"] == "lsd"


def test_merges_and_n_umis_behind_the_splitter_sort(splitter):
    tp._both_merge(dict(n_cells=30, n_genes=2000, umi_len=12, permille_neighbour=150), 200_000, 3, 20,
                   "10x_aug_2016_split", capi.BARCODES_CONST)
    tp._both_merge(dict(n_cells=25, n_genes=1500, umi_len=8, permille_neighbour=120), 120_000, 3, 10,
                   "indrop_v3", capi.BARCODES_CONST)
    tp.test_reference_fixture_umi_merge_strategy_simple()


@pytest.mark.parametrize("seed", range(8))
def test_random_small_streams(splitter, seed):
    rng = np.random.default_rng(31000 + seed)
    ts.run_case(rng, n=int(rng.integers(512, 9000)), n_cb=int(rng.integers(1, 60)), n_gene=int(rng.integers(1, 40)),
                n_umi=int(rng.integers(1, 80)))


@pytest.mark.parametrize("n", [512, 513, 4095, 4096, 4097, 8193, 16385, 40_000])
def test_tile_boundary_sizes(splitter, n):
    ts.run_case(np.random.default_rng(n), n=n, n_cb=7, n_gene=5, n_umi=9, min_before=0, min_after=0)


def test_buckets_beyond_the_lds_sort_fall_back(splitter):
    """One molecule with 20 000 reads (or only gene-less reads of one barcode) is ONE fine bucket, larger than the LDS sort
    takes: the pass must notice before the second partition and hand over to the LSD sort."""
    rng = np.random.default_rng(5)
    ts.run_case(rng, n=20_000, n_cb=1, n_gene=1, n_umi=1, p_nogene=0.0, min_before=0, min_after=0)
    ts.run_case(rng, n=20_000, n_cb=1, n_gene=1, n_umi=1, p_nogene=1.0, min_before=0, min_after=0)
    ts.run_case(rng, n=12_000, n_cb=12_000, n_gene=3, n_umi=4, cb_len=(16, 16), min_before=0, min_after=0)
    # a hot molecule of ~6 000 reads among others: still inside the LDS sort (512 threads x 16)
    ts.run_case(rng, n=30_000, n_cb=3, n_gene=2, n_umi=2, p_nogene=0.0, min_before=0, min_after=0)


def test_variable_lengths_and_ns(splitter):
    rng = np.random.default_rng(9)
    ts.run_case(rng, n=5000, n_cb=30, n_gene=12, n_umi=40, cb_len=(8, 19), umi_len=(6, 6), n_rate=0.05, min_before=0)
    ts.run_case(rng, n=5000, n_cb=30, n_gene=12, n_umi=25, cb_len=(10, 10), umi_len=(4, 9), min_before=1)
    ts.run_case(rng, n=4000, n_cb=25, n_gene=6, n_umi=12, umi_len=(5, 5), n_rate=0.3, cb_n_rate=0.1, min_before=0)


def _digest(c):
    import hashlib
    h = hashlib.sha256()
    for a in c.molecules():
        h.update(np.ascontiguousarray(a).tobytes())
    for filt in (True, False):
        for a in c.count_matrix_csc(filtered=filt):
            h.update(np.ascontiguousarray(a).tobytes())
    h.update(np.ascontiguousarray(c.cell_rows()).tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("shape", ["c2", "c3"])
def test_splitter_equals_lsd_at_baseline_size(monkeypatch, shape):
    """1e8 reads (C2: keys only, 57-bit key; C3 shape: 64-bit key + mark byte, with the whitelist merge): the splitter path
    and the LSD path must produce the same molecule table, cell rows and matrices, byte for byte."""
    if shape == "c2":
        s = SynthStream(n_reads=100_000_000, n_cells=5000, n_genes=30000)
        kw = dict(min_genes_before_merge=20, min_genes_after_merge=100)
    else:
        s = SynthStream(n_reads=100_000_000, n_cells=20000, n_genes=30000, umi_len=12, stream_id=3)
        kw = dict(merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=capi.BARCODES_CONST,
                  barcodes_file=os.path.join(tp.DATA, "10x_aug_2016_split"), min_genes_before_merge=20, min_genes_after_merge=100)
    dev = s.generate_device(0)
    c = capi.Context(**kw)
    c.push_reads_device(*dev.ptrs, dev.n, adopt=True)
    digests = {}
    for mode in ("lsd", "splitter"):
        monkeypatch.setenv("DROPEST_SORT", mode)
        c.reset_results(); c.set_initialized(); c.merge_and_filter()
        assert c.sort_layout()["sort"] == mode
        digests[mode] = _digest(c)
    assert digests["lsd"] == digests["splitter"]
    dev.free()


def test_a_bucket_left_unsorted_is_caught_and_the_lsd_sort_takes_over(splitter, monkeypatch):
    """DROPEST_SS_DEBUG=2 makes ss_local skip its sort passes: every bucket reaches the head detection unsorted.  The in-kernel
    order check (one compare per record, every pass) must raise the device flag, the host must rebuild the keys and hand the pass
    to the LSD sort -- results equal the oracle's, and the violation is counted.  (What the check guards: the one-atomic ranking
    of ss_local leans on the LDS applying same-address lanes of one instruction in lane order.)"""
    monkeypatch.setenv("DROPEST_SS_DEBUG", "2")
    o, c = tp._both(dict(n_cells=60, n_genes=3000), 300_000, 10, 30)
    assert c.sort_layout()["sort"] == "lsd"
    monkeypatch.delenv("DROPEST_SS_DEBUG")
    o, c = tp._both(dict(n_cells=60, n_genes=3000), 300_000, 10, 30)
    assert c.sort_layout()["sort"] == "splitter"


# ---- partitions by reservation (k_ssort.h: no histogram passes; every bucket a region of fixed capacity) -----------------------------
def test_reservation_equals_the_counting_partitions(monkeypatch):
    """2e7 reads of the C2 shape and of the C3 shape (mark byte beside the key): the same pass with the partitions placing their records by
    reservation (the default) and by counting first (DROPEST_SS_NO_RESERVE=1) gives the same molecule table, cell rows and matrices,
    byte for byte; the reservation pass runs no histogram kernel."""
    for kw_s, force_byte in ((dict(n_cells=1000, n_genes=30000), False), (dict(n_cells=4000, n_genes=30000, umi_len=12, stream_id=3), True)):
        if force_byte:
            monkeypatch.setenv("DROPEST_FORCE_BYTE_VALUES", "1")
        s = SynthStream(n_reads=20_000_000, **kw_s)
        dev = s.generate_device(0)
        c = capi.Context(min_genes_before_merge=20, min_genes_after_merge=100)
        c.push_reads_device(*dev.ptrs, dev.n, adopt=True)
        digests = {}
        for mode in ("reserve", "count"):
            if mode == "count":
                monkeypatch.setenv("DROPEST_SS_NO_RESERVE", "1")
            c.set_profiling(True)
            c.reset_results(); c.set_initialized(); c.merge_and_filter()
            assert c.sort_layout()["sort"] == "splitter" and c.sort_layout()["value_bytes"] == (1 if force_byte else 0)
            names = set(c.kernel_stats())
            assert ("ss_hist:L1" in names) == (mode == "count") and "count:ss_reserve_overflow" not in names, names
            digests[mode] = _digest(c)
            c.set_profiling(False)
        monkeypatch.delenv("DROPEST_SS_NO_RESERVE")
        assert digests["reserve"] == digests["count"]
        c.close(); dev.free()
        monkeypatch.delenv("DROPEST_FORCE_BYTE_VALUES", raising=False)


def test_a_region_that_overflows_is_noticed_and_the_counting_partitions_take_over(splitter, monkeypatch):
    """Regions of 101 % of the mean bucket size overflow on any stream (fine buckets scatter by 12 %): the flag must come back, the keys
    must be rebuilt and the counting partitions must finish the pass -- results equal the oracle's, the overflow is counted."""
    monkeypatch.setenv("DROPEST_SS_CAP2_PERCENT", "101")
    cb, umi, gene, aux = parity.canonical_stream(*SynthStream(n_reads=400_000, n_cells=60, n_genes=3000).generate_host())
    o = parity.oracle_run(Oracle, dict(merge_kind=0, min_genes_before=10, min_genes_after=30), cb, umi, gene, aux)
    c = capi.Context(min_genes_before_merge=10, min_genes_after_merge=30)
    c.set_profiling(True)
    c.push_reads(cb, umi, gene, aux)
    c.set_initialized(); c.merge_and_filter()
    parity.compare(o, c)
    st = c.kernel_stats()
    assert c.sort_layout()["sort"] == "splitter" and "count:ss_reserve_overflow" in st and "ss_hist:L1" in st
    c.close()
