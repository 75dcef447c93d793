"""The sharded runner against the ORACLE directly (tests/test_gpu_multi.py compares shards with one context, which other tests pin on the
oracle on other streams: transitive).  One case per merge kind, N-UMIs and -u included; shards whose ordinal ranges start beyond 2^32
(BASELINE configs[4]: 8e9 reads over 8 GPUs -- ranks 4..7 start at ordinals >= 4.3e9); a mixed exact_widths option."""
import os

import numpy as np
import pytest

from dropest_amd import capi
from dropest_amd.multi import ShardGroup
from dropest_amd.synth import SynthStream, inject_n
from oracle import Oracle

import parity

pytestmark = pytest.mark.gpu

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dropest_amd", "data", "barcodes")


def oracle_kw(kw):
    """capi.make_cfg keyword arguments -> oracle.OracleConfig keyword arguments (same defaults on both sides)."""
    names = {"merge_kind": "merge_kind", "barcodes_kind": "barcodes_kind", "barcodes_file": "barcodes_file",
             "min_genes_before_merge": "min_genes_before", "min_genes_after_merge": "min_genes_after", "min_merge_fraction": "min_merge_fraction",
             "max_cb_merge_edit_distance": "max_cb_merge_ed", "umi_merge_kind": "umi_merge_kind", "max_umi_merge_edit_distance": "max_umi_merge_ed",
             "umi_merge_multiplier": "umi_mult", "gene_match_levels": "match_levels", "max_cells": "max_cells", "max_merge_prob": "max_merge_prob",
             "max_real_merge_prob": "max_real_merge_prob"}
    out = {names[k]: v for k, v in kw.items()}
    out.setdefault("max_cb_merge_ed", 2)            # (make_cfg's default)
    return out


def run_shards(arrays, kw, bounds, first_ordinals=None, side=(), options=(), steps=1):
    """Shard i gets reads [bounds[i], bounds[i + 1]) of the stream as the ordinal range that starts at first_ordinals[i] (default: bounds[i])."""
    world = len(bounds) - 1
    g = ShardGroup([0] * world, **kw)
    for i, s in enumerate(g.shards):
        if side:
            s.set_side_strings(side)
        for k, v in (options[i] if options else {}).items():
            s.set_option(k, v)
        s.set_reads(capi.DeviceArrays.from_host(0, *[a[bounds[i]:bounds[i + 1]] for a in arrays]), int(first_ordinals[i]) if first_ordinals is not None else bounds[i])
    for _ in range(steps):
        g.step()
    s0 = g.shards[0]
    out = {"cm": [x.copy() for x in s0.matrix(True)], "raw": [x.copy() for x in s0.matrix(False)], "merged": s0.merged_barcodes(),
           "form": (s0.matrix_form(True), s0.matrix_form(False)), "phases": s0.phase_stats()}
    g.close()
    return out


def check_vs_oracle(got, o, side=()):
    """Both global matrices entry for entry, their column barcodes (cm: the filtered cells in compare_cells order, cm_raw: the real cells in
    cell-id order) and the merged (source -> target) barcodes against the oracle's container."""
    orows = o.cell_rows()    # merged, excluded, real, ...
    for filt, name in ((True, "cm"), (False, "raw")):
        og, ocol, ov = o.count_matrix(filtered=filt)
        p, i, x, b = got[name]
        ncols = len(p) - 1
        assert len(i) == len(og), "%s: nnz %d vs %d" % (name, len(i), len(og))
        col = np.repeat(np.arange(ncols, dtype=np.uint64), np.diff(p.astype(np.int64)))
        assert np.array_equal(i.astype(np.uint64), og) and np.array_equal(col, ocol) and np.array_equal(x.astype(np.uint64), ov), name
        cells = list(o.filtered_cells()) if filt else [int(k) for k in np.flatnonzero(orows[:, 2] != 0)]
        assert [capi.unpack_code(int(c), side) for c in b] == [o.cell_barcode(int(k)) for k in cells], name
    mt = np.asarray(o.merge_targets(), np.int64)
    want = {o.cell_barcode(int(k)): o.cell_barcode(int(mt[k])) for k in np.flatnonzero(mt != np.arange(len(mt)))}
    have = {capi.unpack_code(int(a), side): capi.unpack_code(int(b), side) for a, b in zip(*got["merged"])}
    assert have == want
    return want


def even_bounds(n, world):
    return [n * i // world for i in range(world + 1)]


WL_10X = {"barcodes_kind": capi.BARCODES_CONST, "barcodes_file": os.path.join(DATA, "10x_aug_2016_split")}
WL_INDROP = {"barcodes_kind": capi.BARCODES_CONST, "barcodes_file": os.path.join(DATA, "indrop_v3")}
KINDS = {
    # name: (stream, container configuration, N-UMI rate)
    "none": (dict(n_reads=300_000, n_cells=50, n_genes=3000), dict(min_genes_before_merge=10, min_genes_after_merge=30), 0.0),
    "none+N": (dict(n_reads=200_000, n_cells=40, n_genes=1500, umi_len=8), dict(min_genes_before_merge=10, min_genes_after_merge=20), 1e-2),
    "real:10x": (dict(n_reads=300_000, n_cells=40, n_genes=2000, umi_len=12, permille_neighbour=150),
                 dict(merge_kind=capi.MERGE_REAL_BARCODES, min_genes_before_merge=3, min_genes_after_merge=20, **WL_10X), 0.0),
    "real:indrop+N": (dict(n_reads=300_000, n_cells=40, n_genes=2000, umi_len=8, permille_neighbour=150, whitelist="indrop_v3"),
                      dict(merge_kind=capi.MERGE_REAL_BARCODES, min_genes_before_merge=3, min_genes_after_merge=20, **WL_INDROP), 5e-3),
    "poisson_real": (dict(n_reads=300_000, n_cells=40, n_genes=2000, umi_len=12, permille_neighbour=150),
                     dict(merge_kind=capi.MERGE_POISSON_REAL, min_genes_before_merge=3, min_genes_after_merge=20, **WL_10X), 0.0),
    "simple": (dict(n_reads=150_000, n_cells=25, n_genes=1200, umi_len=8, permille_neighbour=150),
               dict(merge_kind=capi.MERGE_SIMPLE, max_cb_merge_edit_distance=2, min_merge_fraction=0.2, min_genes_before_merge=3, min_genes_after_merge=10), 0.0),
    "simple:ties": (dict(n_reads=60_000, n_cells=12, n_genes=40, umi_len=3, permille_neighbour=250),
                    dict(merge_kind=capi.MERGE_SIMPLE, max_cb_merge_edit_distance=17, min_merge_fraction=0.0, min_genes_before_merge=1, min_genes_after_merge=1), 0.0),
    "poisson_simple": (dict(n_reads=150_000, whitelist="10x_aug_2016_split", n_cells=30, n_genes=1500, umi_len=8, permille_neighbour=150),
                       dict(merge_kind=capi.MERGE_POISSON_SIMPLE, max_cb_merge_edit_distance=1, max_real_merge_prob=1e-3, min_genes_before_merge=3,
                            min_genes_after_merge=10), 0.0),
    "all": (dict(n_reads=150_000, whitelist="10x_aug_2016_split", n_cells=30, n_genes=1500, umi_len=10, permille_neighbour=150),
            dict(merge_kind=capi.MERGE_ALL, max_cb_merge_edit_distance=2, min_genes_before_merge=3, min_genes_after_merge=10), 0.0),
    "directional+N": (dict(n_reads=150_000, n_cells=30, n_genes=400, umi_len=6, reads_per_molecule=3),
                      dict(min_genes_before_merge=5, min_genes_after_merge=10, umi_merge_kind=capi.UMI_MERGE_DIRECTIONAL, max_umi_merge_edit_distance=1,
                           umi_merge_multiplier=2.0), 2e-2),
    "real+directional": (dict(n_reads=200_000, n_cells=40, n_genes=600, umi_len=8, permille_neighbour=150, reads_per_molecule=3),
                         dict(merge_kind=capi.MERGE_REAL_BARCODES, min_genes_before_merge=3, min_genes_after_merge=10, umi_merge_kind=capi.UMI_MERGE_DIRECTIONAL,
                              max_umi_merge_edit_distance=1, umi_merge_multiplier=2.0, **WL_10X), 0.0),
}


def seeded_oracle(kw, arrays, side):
    """The directional UMI merge never seeds rand() (its random fills of N-UMIs continue the PROCESS's sequence): the oracle -- which
    draws from the C library -- starts from srand(1) like a fresh reference process; the library restates that sequence per pass."""
    import ctypes
    ctypes.CDLL("libc.so.6").srand(1)
    return parity.oracle_run(Oracle, oracle_kw(kw), *arrays, side)


def make_case(name):
    stream_kw, kw, n_rate = KINDS[name]
    cb, umi, gene, aux = parity.canonical_stream(*SynthStream(**stream_kw).generate_host())
    side = ()
    if n_rate:
        umi, side = inject_n(umi, gene, n_rate, 7, stream_kw.get("umi_len", 10))
        assert len(side) > 10
    return (cb, umi, gene, aux), kw, side


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", sorted(KINDS))
def test_shards_match_the_oracle(name, world):
    """Every merge kind (and the UMI merges) through `world` shards against the oracle's container over the whole stream."""
    arrays, kw, side = make_case(name)
    o = seeded_oracle(kw, arrays, side)
    got = run_shards(arrays, kw, even_bounds(len(arrays[0]), world), side=side)
    want = check_vs_oracle(got, o, side)
    assert len(got["cm"][3]) >= 8
    if kw.get("merge_kind", 0):
        assert len(want) > 10                   # the merge really merged
    assert got["form"] == (3, 3)                # the step ended with the 32-bit slots (widened from the byte form by the shards' host threads)


@pytest.mark.parametrize("name", ["none+N", "real:indrop+N", "simple:ties", "directional+N"])
def test_eight_shards_with_ordinal_ranges_beyond_2_to_32(name):
    """BASELINE configs[4] is 8e9 reads over 8 GPUs: ranks 4..7 hold ordinals beyond 2^32.  Eight shards whose ranges start at i x 1.0e9
    (ragged lengths) must give what ONE container gives over the concatenation -- first-seen cell ids (CellsDataContainer.cpp:64), the
    global UMI order and the one rand() sequence of the N-UMI fills (MergeUMIsStrategySimple.cpp:34-41, 76-88) all hang on the ordinals --
    and what the oracle gives."""
    arrays, kw, side = make_case(name)
    n = len(arrays[0])
    rng = np.random.default_rng(5)
    cuts = np.sort(rng.choice(np.arange(1, n), 7, replace=False))
    bounds = [0] + [int(c) for c in cuts] + [n]
    firsts = [i * 1_000_000_000 for i in range(8)]
    assert firsts[5] > 2 ** 32
    o = seeded_oracle(kw, arrays, side)
    got = run_shards(arrays, kw, bounds, first_ordinals=firsts, side=side, steps=2)
    check_vs_oracle(got, o, side)
    plain = run_shards(arrays, kw, bounds, side=side)          # the same shards with ordinals below 2^32: the same matrices
    for k in ("cm", "raw"):
        assert all(np.array_equal(a, b) for a, b in zip(got[k], plain[k]))


def test_shards_that_disagree_on_exact_widths():
    """one shard told to measure the exchange record's widths exactly, the other sampling them -- they must still lay the record
    out alike and enter the same collectives (a hang or corrupted reads before)."""
    arrays, kw, side = make_case("real:10x")
    arrays = [a.copy() for a in arrays]
    arrays[3][512 * 3 + 17] = (int(arrays[3][512 * 3 + 17]) & 0xFFFF0000) | 0x1FFF       # a chromosome id the sample does not see: the partition is repeated
    o = seeded_oracle(kw, arrays, side)
    for opts in ([{"exact_widths": 1}, {}], [{}, {"exact_widths": 1}, {}]):
        got = run_shards(arrays, kw, even_bounds(len(arrays[0]), len(opts)), side=side, options=opts, steps=2)
        check_vs_oracle(got, o, side)


def test_slots_off_ends_at_the_byte_form_and_widens_to_the_same_matrix():
    arrays, kw, side = make_case("real:10x")
    bounds = even_bounds(len(arrays[0]), 3)
    on = run_shards(arrays, kw, bounds)
    off = run_shards(arrays, kw, bounds, options=[{"slots_matrix": 0}] * 3)
    assert on["form"] == (3, 3) and off["form"] == (2, 2)
    for k in ("cm", "raw"):
        assert all(np.array_equal(a, b) for a, b in zip(on[k], off[k]))


def test_an_empty_filtered_matrix_in_a_slots_step():
    """No cell passes min_genes_after_merge: cm has no column and no entry, cm_raw is as ever (the shared slots buffer of an empty matrix
    used to be asked for with zero bytes: found by scripts/soak_sharded.py)."""
    arrays, kw, side = make_case("none")
    kw = dict(kw, min_genes_after_merge=100_000)
    o = seeded_oracle(kw, arrays, side)
    got = run_shards(arrays, kw, even_bounds(len(arrays[0]), 3), steps=2)
    check_vs_oracle(got, o, side)
    assert len(got["cm"][1]) == 0 and len(got["cm"][3]) == 0 and len(got["raw"][1]) > 1000


@pytest.mark.parametrize("slots", [0, 1])
def test_host_planned_cm_raw_beside_cm_with_eight_shards(slots):
    """cm_raw planned on the HOST (raw_on_device 0) goes through the same assembling code as cm, right before it: its column descriptors were
    staged in the buffer cm's were then written into while the copy to the device was still pending (nothing waits between the two since the
    emit stopped waiting) -- wrong entries or a device fault in two runs of five (scripts/soak_sharded.py found it).  Each matrix has its own
    buffers now; repeated here with lists that overflow (the matrix is then assembled twice inside one step)."""
    arrays, kw, side = make_case("real:10x")
    o = seeded_oracle(kw, arrays, side)
    n = len(arrays[0])
    cuts = [0, n // 9, n // 7, n // 5, n // 3, n // 2, 2 * n // 3, 7 * n // 8, n]
    for cap in (0, 16):
        opts = [{"raw_on_device": 0, "slots_matrix": slots, "byte_list_cap": cap, "narrow_matrix": 0}] * 8
        for _ in range(3):
            check_vs_oracle(run_shards(arrays, kw, cuts, options=opts, steps=2), o, side)


def test_cell_id_by_cb_answers_from_the_host_mirror():
    """dropest_cell_id_by_cb (CellsDataContainer::cell_id_by_cb): real cells from the host's rows, any other barcode from one fetch of the
    pass's barcode list -- against the oracle, before and after a second pass."""
    s = SynthStream(n_reads=200_000, n_cells=30, n_genes=1500)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    o = parity.oracle_run(Oracle, dict(min_genes_before=10, min_genes_after=20), cb, umi, gene, aux)
    c = parity.gpu_run(dict(min_genes_before_merge=10, min_genes_after_merge=20), cb, umi, gene, aux)
    rows = c.cell_rows()
    real = np.flatnonzero(rows["is_real"])
    other = np.flatnonzero(rows["is_real"] == 0)[:200]
    assert len(real) > 10 and len(other) > 50
    for rnd in range(2):
        for k in list(real[:40]) + list(other):
            assert c.cell_id_by_cb(int(rows["barcode"][k])) == int(k) == o.cell_id_by_cb(capi.unpack_code(int(rows["barcode"][k])))
        assert c.cell_id_by_cb(int(capi.pack_seq("TTTTTTTTTTTTTTTT"))) == -1
        c.reset_results(); c.set_initialized(); c.merge_and_filter()


def test_a_shard_whose_byte_lists_overflow_places_its_columns_as_32_bit_slots(monkeypatch):
    """A row list too short for a shard's sparse columns (DROPEST_MATRIX_ROW_LIST_CAP: a dozen entries): that shard's widening reports the
    overflow and the shard places the 32-bit form of its columns into the shared slots itself -- same matrices, no collective decision."""
    arrays = parity.canonical_stream(*SynthStream(n_reads=200_000, n_cells=50, n_genes=30000).generate_host())   # 30 000 genes: small cells list most rows
    kw, side = dict(min_genes_before_merge=10, min_genes_after_merge=30), ()
    bounds = even_bounds(len(arrays[0]), 3)
    o = seeded_oracle(kw, arrays, side)
    want = run_shards(arrays, kw, bounds)
    check_vs_oracle(want, o, side)                      # (many listed rows: the lists' local positions are translated to global places)
    assert int((np.diff(want["raw"][1].astype(np.int64)) >= 255).sum()) > 300
    monkeypatch.setenv("DROPEST_MATRIX_ROW_LIST_CAP", "12")
    got = run_shards(arrays, kw, bounds, steps=2)
    assert got["form"] == (3, 3) and got["phases"]["matrix:overflow"]["steps"] >= 2
    check_vs_oracle(got, o, side)


@pytest.mark.parametrize("world,chunks", [(1, 4), (2, 2), (3, 5), (8, 3)])
@pytest.mark.parametrize("name", ["none", "none+N", "real:10x", "real:indrop+N"])
def test_chunked_exchange_matches_the_oracle(name, world, chunks, monkeypatch):
    """The all-to-all leaves chunk by chunk under the partition, and the barcode table is built chunk by chunk as the pieces land (shard
    option exchange_chunks; on by itself from 2^22 reads per shard): ranges of the same read arrays, so every read keeps its first-seen
    ordinal -- the hot list and the sampled table sizing (forced on these small streams) see the first chunk only."""
    monkeypatch.setenv("DROPEST_CB_SAMPLE_MIN", "1000")
    arrays, kw, side = make_case(name)
    o = seeded_oracle(kw, arrays, side)
    opts = [{"exchange_chunks": chunks, "force_exchange": 1}] * world
    got = run_shards(arrays, kw, even_bounds(len(arrays[0]), world), side=side, options=opts, steps=2)
    check_vs_oracle(got, o, side)
    if "+N" not in name:    # (a UMI with N is an escaped code with bit 63 set: those streams travel as five arrays, in one piece)
        assert got["phases"]["all_to_all:chunks"]["steps"] == 2 * chunks


# ---- the UMI dictionary across shards (round 6; Estimation/StringIndexer.cpp:10-18 has no width limit) ----
def _dictionary_phase(got):
    return got["phases"].get("umi_dictionary", {"steps": 0})["steps"]


@pytest.mark.parametrize("world,n_rate", [(3, 0.0), (3, 5e-3), (2, 0.0), (8, 2e-3)])
def test_gene_and_umi_of_64_bits_over_shards(world, n_rate):
    """30-base UMIs (61 bits with the sentinel) beside a few thousand genes: gene + UMI alone pass 64 bits.  One context keys the UMIs by their
    rank in a dictionary since round 5; the shards of a sharded run used to refuse.  Now every shard's distinct UMIs are gathered, the union is
    sorted once per shard, and the ranks -- ascending with the codes, the same everywhere -- key the molecules: against the oracle, with N-UMIs
    (random fills against the one global rand() sequence) and two passes on the same shards."""
    s = SynthStream(n_reads=120_000, n_cells=30, n_genes=2500, umi_len=30)
    arrays = parity.canonical_stream(*s.generate_host())
    side = ()
    if n_rate:
        umi, side = inject_n(arrays[1], arrays[2], n_rate, 7, 30)
        arrays = (arrays[0], umi, arrays[2], arrays[3])
    kw = dict(min_genes_before_merge=5, min_genes_after_merge=15)
    o = seeded_oracle(kw, arrays, side)
    got = run_shards(arrays, kw, even_bounds(len(arrays[0]), world), side=side, steps=2)
    assert _dictionary_phase(got) >= 2
    check_vs_oracle(got, o, side)
    assert len(got["cm"][3]) >= 8


def test_wide_umis_whitelist_merge_across_shards():
    """... and with -m: molecule rows cross between the shards carrying ranks of the one dictionary."""
    s = SynthStream(n_reads=200_000, n_cells=40, n_genes=2000, umi_len=30, permille_neighbour=150)
    arrays = parity.canonical_stream(*s.generate_host())
    umi, side = inject_n(arrays[1], arrays[2], 2e-3, 5, 30)
    arrays = (arrays[0], umi, arrays[2], arrays[3])
    kw = dict(merge_kind=capi.MERGE_REAL_BARCODES, min_genes_before_merge=3, min_genes_after_merge=20, **WL_10X)
    o = seeded_oracle(kw, arrays, side)
    got = run_shards(arrays, kw, even_bounds(len(arrays[0]), 3), side=side)
    assert _dictionary_phase(got) >= 1
    want = check_vs_oracle(got, o, side)
    assert len(want) > 10


@pytest.mark.parametrize("name", sorted(KINDS))
def test_every_merge_kind_over_shards_with_the_dictionary_forced(name, monkeypatch):
    """The suite's own sharded cases once more with DROPEST_UMI_DICT=2: every consumer of the key's UMI field across shards (N-UMI groups, -u,
    the UMI-gene index of the whitelist-free merges, the Poisson estimators' UMI histograms, imported molecule rows) must cope with ranks."""
    monkeypatch.setenv("DROPEST_UMI_DICT", "2")
    arrays, kw, side = make_case(name)
    o = seeded_oracle(kw, arrays, side)
    got = run_shards(arrays, kw, even_bounds(len(arrays[0]), 3), side=side)
    assert _dictionary_phase(got) >= 1
    check_vs_oracle(got, o, side)
