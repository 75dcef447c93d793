"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same seeded streams.
Integer / index work: the bar is bit-exact equality of every observable."""
import numpy as np
import pytest

from dropest_amd import capi
from dropest_amd.synth import SynthStream
from oracle import Oracle

import parity

pytestmark = pytest.mark.gpu


def _both(stream_kw, n_reads, min_before, min_after, chunks=1, max_cells=-1, levels="eEBA", reads_output=False):
    s = SynthStream(n_reads=n_reads, **stream_kw)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    o = parity.oracle_run(Oracle, dict(merge_kind=0, min_genes_before=min_before, min_genes_after=min_after,
                                       match_levels=levels, max_cells=max_cells), cb, umi, gene, aux)
    c = parity.gpu_run(dict(merge_kind=capi.MERGE_NONE, min_genes_before_merge=min_before,
                            min_genes_after_merge=min_after, gene_match_levels=levels, max_cells=max_cells),
                       cb, umi, gene, aux, chunks=chunks)
    parity.compare(o, c, reads_output=reads_output)
    return o, c


def test_tiny_stream():
    _both(dict(n_cells=8, n_genes=50, umi_len=6), 2000, 3, 5)


def test_c2_shape_100k():
    """10x v2 shape (CB 16 + UMI 10), no CB merge -- the C2 configuration scaled down."""
    o, c = _both(dict(n_cells=40, n_genes=3000), 100_000, 20, 100)
    assert len(c.filtered_cells()) > 10


def test_c2_shape_1m_chunked():
    _both(dict(n_cells=200, n_genes=8000), 1_000_000, 20, 100, chunks=7)


def test_c2_shape_6m_at_natural_thresholds():
    """Past 2^22 reads one context switches, without any environment override, to the barcode table sized from a sample, the LDS
    table of hot barcodes, the key layout planned from a sample (exact statistics gathered with the keys) and the splitter sort with
    its one-atomic ranking: every observable against the oracle (~15 s of oracle time)."""
    o, c = _both(dict(n_cells=300, n_genes=30000), 6_000_000, 20, 100, chunks=3)
    assert len(c.filtered_cells()) > 250
    lay = c.sort_layout()
    assert lay["sort"] == "splitter"


def test_c3_shape_5m_with_n_umis_and_merge_at_natural_thresholds():
    s = SynthStream(n_reads=5_000_000, n_cells=400, n_genes=20000, umi_len=12, permille_neighbour=100, stream_id=3)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    from dropest_amd.synth import inject_n
    umi, side = inject_n(umi, gene, 2e-4, 3, 12)
    wl = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dropest_amd", "data", "barcodes", "10x_aug_2016_split")
    o = parity.oracle_run(Oracle, dict(merge_kind=1, barcodes_kind=capi.BARCODES_CONST, barcodes_file=wl, min_genes_before=10, min_genes_after=50),
                          cb, umi, gene, aux, side)
    c = parity.gpu_run(dict(merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=capi.BARCODES_CONST, barcodes_file=wl, min_genes_before_merge=10,
                            min_genes_after_merge=50), cb, umi, gene, aux, side)
    parity.compare(o, c, side)
    assert int(c.cell_rows()["is_merged"].sum()) > 1000


def test_context_reused_for_another_stream_equals_a_fresh_context():
    """dropest_clear_reads + a different stream on the same context: its buffers hold what the previous stream left (molecule
    tables, barcode table, sort scratch of another size), the result must be that of a fresh context."""
    wl = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dropest_amd", "data", "barcodes", "10x_aug_2016_split")
    kw = dict(min_genes_before_merge=10, min_genes_after_merge=60, merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=capi.BARCODES_CONST, barcodes_file=wl)

    def outputs(c):
        rows = c.cell_rows()
        return [x.copy() for x in c.count_matrix_csc(filtered=True)] + [x.copy() for x in c.count_matrix_csc(filtered=False)] + \
               [rows[k].copy() for k in rows.dtype.names] + [np.array(c.merge_targets()), np.array(c.filtered_cells())]
    shapes = [dict(n_reads=24_000_000, n_cells=6000, n_genes=20000, umi_len=12, stream_id=11, permille_neighbour=120),
              dict(n_reads=9_000_000, n_cells=900, n_genes=4000, umi_len=8, stream_id=12, permille_neighbour=60),
              dict(n_reads=17_000_000, n_cells=15000, n_genes=30000, umi_len=10, stream_id=13, permille_neighbour=180)]
    c = capi.Context(**kw)
    for shape in shapes:
        dev = SynthStream(**shape).generate_device(0)
        c.clear_reads()
        c.push_reads_device(*dev.ptrs, dev.n, adopt=True)
        c.set_initialized(); c.merge_and_filter()
        a = outputs(c)
        f = capi.Context(**kw)
        f.push_reads_device(*dev.ptrs, dev.n, adopt=True)
        f.set_initialized(); f.merge_and_filter()
        b = outputs(f)
        f.close()
        assert len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b)), shape
        c.clear_reads()
        dev.free()
    c.close()


def test_query_levels_and_reads_output():
    _both(dict(n_cells=30, n_genes=1500), 60_000, 10, 10, levels="e", reads_output=True)
    _both(dict(n_cells=30, n_genes=1500), 60_000, 10, 10, levels="iI")


def test_max_cells_cut():
    o, c = _both(dict(n_cells=40, n_genes=3000), 80_000, 10, 10, max_cells=7)
    assert len(c.filtered_cells()) == 7


def test_v3_shape_umi12():
    _both(dict(n_cells=60, n_genes=4000, umi_len=12), 150_000, 20, 50)


def test_single_read_and_empty():
    cb = np.array([capi.pack_seq("ACGTACGTACGTACGT")], np.uint64)
    umi = np.array([capi.pack_seq("ACGTACGTAC")], np.uint64)
    gene = np.array([0], np.uint32); aux = np.array([0 | (2 << 16)], np.uint32)
    o = parity.oracle_run(Oracle, dict(min_genes_before=0, min_genes_after=0), cb, umi, gene, aux)
    c = parity.gpu_run(dict(min_genes_before_merge=0, min_genes_after_merge=0), cb, umi, gene, aux)
    parity.compare(o, c)
    e = capi.Context(min_genes_before_merge=0, min_genes_after_merge=0)
    e.set_initialized(); e.merge_and_filter()
    assert e.total_cells_number() == 0 and len(e.filtered_cells()) == 0
    assert len(e.count_matrix()[0]) == 0


def test_intergenic_only_cell_and_ragged():
    """A barcode with only gene-less reads still becomes a cell (CellsDataContainer.cpp:64-69)."""
    P = capi.pack_seq
    recs = [("AAAACCCCGGGGTTTT", "AAAAAAAAAA", None, 1, 2), ("AAAACCCCGGGGTTTT", "AAAAAAAAAC", None, 2, 2),
            ("CCCCAAAAGGGGTTTT", "AAAAAAAAAA", 0, 0, 2), ("CCCCAAAAGGGGTTTT", "AAAAAAAAAA", 0, 0, 4),
            ("CCCCAAAAGGGGTTTT", "AAAAAAAAAA", 1, 1, 1), ("GGGGAAAACCCCTTTT", "TTTTTTTTTT", 1, 1, 3)]
    cb = np.array([P(r[0]) for r in recs], np.uint64); umi = np.array([P(r[1]) for r in recs], np.uint64)
    gene = np.array([capi.NO_GENE if r[2] is None else r[2] for r in recs], np.uint32)
    aux = np.array([r[3] | (r[4] << 16) for r in recs], np.uint32)
    cb, umi, gene, aux = parity.canonical_stream(cb, umi, gene, aux)
    o = parity.oracle_run(Oracle, dict(min_genes_before=0, min_genes_after=0), cb, umi, gene, aux)
    c = parity.gpu_run(dict(min_genes_before_merge=0, min_genes_after_merge=0), cb, umi, gene, aux)
    parity.compare(o, c)
    assert c.total_cells_number() == 3


def test_state_machine_errors():
    """add_record after set_initialized / merge before init are errors (CellsDataContainer.cpp:41-42, :61-62, :165-166)."""
    c = capi.Context(min_genes_before_merge=0, min_genes_after_merge=0)
    with pytest.raises(capi.DropestError):
        c.merge_and_filter()
    cb = np.array([capi.pack_seq("ACGT")], np.uint64)
    c.push_reads(cb, cb, np.array([0], np.uint32), np.array([2 << 16], np.uint32))
    c.set_initialized()
    with pytest.raises(capi.DropestError):
        c.set_initialized()
    with pytest.raises(capi.DropestError):
        c.push_reads(cb, cb, np.array([0], np.uint32), np.array([2 << 16], np.uint32))


def test_device_generator_matches_host():
    s = SynthStream(n_reads=300_000, n_cells=100, n_genes=5000)
    host = s.generate_host(first=1000, n=50_000)
    dev = s.generate_device(0, first=1000, n=50_000)
    got = dev.to_host()
    dev.free()
    for a, b in zip(host, got):
        assert np.array_equal(a, b)


def test_sortedness_and_checksum_at_scale():
    """BASELINE configs[1] at FULL size (1e8 reads, 5000 cells): size-independent properties on a stream far too
    long for the oracle -- conservation of reads, strict sortedness (no duplicate molecule), per-cell sums."""
    s = SynthStream(n_reads=100_000_000, n_cells=5000, n_genes=30000)
    dev = s.generate_device(0)
    c = capi.Context(min_genes_before_merge=20, min_genes_after_merge=100)
    c.push_reads_device(*dev.ptrs, dev.n, adopt=True)
    c.set_initialized(); c.merge_and_filter()
    L = c.sort_layout()
    width = L["cell_bits"] + L["gene_bits"] + L["umi_bits"]
    assert width == 57 and (L["passes"], L["sort"]) in ((7, "lsd"), (3, "splitter"))   # LSD: 6 x 8 bits + one 9-bit digit on top
    cell, gene, umi, reads, mark = c.molecules()
    counters = c.global_counters()
    assert int(reads.sum()) + int(counters[0]) == dev.n                      # every read counted exactly once
    key = (cell.astype(np.uint64) << np.uint64(40)) | (gene.astype(np.uint64) << np.uint64(22)) | (umi & np.uint64((1 << 20) - 1))
    assert np.all(key[1:] > key[:-1])                                        # strictly ascending, no duplicate molecule
    rows = c.cell_rows()
    assert int(rows["total_reads"].astype(np.int64).sum()) == int(reads.sum())
    assert int(rows["total_umis"].astype(np.int64).sum()) == len(reads)
    p, i, v = c.count_matrix_csc(filtered=False)
    real = rows["is_real"].astype(bool)
    assert int(v.astype(np.int64).sum()) == int(rows["total_umis"][real].astype(np.int64).sum())
    assert len(p) - 1 == int(real.sum()) and np.all(np.diff(p.astype(np.int64)) == rows["n_genes"][real])
    # the filtered matrix: columns ascending in the compare_cells key, every column has >= min_genes_after_merge genes
    p, i, v = c.count_matrix_csc(filtered=True)
    f = c.filtered_cells().astype(np.int64)
    key = list(zip(rows["requested_genes"][f].tolist(), rows["requested_umis"][f].tolist(), rows["total_umis"][f].tolist()))
    assert key == sorted(key) and min(k[0] for k in key) >= 100
    assert np.all(np.diff(p.astype(np.int64)) == rows["requested_genes"][f])
    # idempotence: a second pass over the same resident stream reproduces every byte
    i1, v1 = i.copy(), v.copy()
    c.reset_results(); c.set_initialized(); c.merge_and_filter()
    p2, i2, v2 = c.count_matrix_csc(filtered=True)
    assert np.array_equal(p, p2) and np.array_equal(i1, i2) and np.array_equal(v1, v2)
    _wire_equals_direct(c)
    dev.free()


def _wire_equals_direct(c):
    """Both matrices as dropest_count_matrix_csc hands them out by default at this size -- bytes over PCIe, widened by host threads under
    the copy (csrc/matrix_decode.h) -- against the same call with the 32-bit arrays copied as they are, and against the public byte form
    decoded by dropest_matrix_bytes_widen: the form bench.py times, verified at the size bench.py runs."""
    for filt in (False, True):
        c.set_matrix_wire(True)
        wire = [x.copy() for x in c.count_matrix_csc(filtered=filt)]
        m = c.count_matrix_csc_bytes(filtered=filt)
        listed = int(m.n_row_listed)
        byt = [x.copy() for x in c.widen_bytes(m)]
        c.set_matrix_wire(False)
        direct = c.count_matrix_csc(filtered=filt)
        assert len(direct[1]) > 1_000_000
        for a, b, d in zip(wire, byt, direct):
            assert np.array_equal(a, d) and np.array_equal(b, d), (filt, listed)
        c.set_matrix_wire(True)
        del wire, byt, direct
    # round 6: with cm_raw's prefetch under way cm crosses the link as one byte per entry of cm_raw and takes its rows from cm_raw's deltas
    # (from 2^26 entries on: the 1e9-read workloads) -- the same slots as the direct copy
    c.set_matrix_wire(True)
    c.prefetch_raw_matrix(form=0)
    cm = [x.copy() for x in c.count_matrix_csc(filtered=True)]
    raw = [x.copy() for x in c.count_matrix_csc(filtered=False)]
    c.set_matrix_wire(False)
    for got, filt in ((cm, True), (raw, False)):
        for a, d in zip(got, c.count_matrix_csc(filtered=filt)):
            assert np.array_equal(a, d), filt
    c.set_matrix_wire(True)


def test_merge_properties_at_scale():
    """C3 shape (UMI 12, Hamming-1 neighbour barcodes, -m + whitelist) at 1e8 reads: properties of the merge."""
    s = SynthStream(n_reads=100_000_000, n_cells=20000, n_genes=30000, umi_len=12, stream_id=3)
    dev = s.generate_device(0)
    c = capi.Context(merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=capi.BARCODES_CONST,
                     barcodes_file=os.path.join(DATA, "10x_aug_2016_split"), min_genes_before_merge=20,
                     min_genes_after_merge=100)
    c.push_reads_device(*dev.ptrs, dev.n, adopt=True)
    c.set_initialized()
    n_real_before = c.real_cells_number()
    c.merge_and_filter()
    rows = c.cell_rows()
    mt = c.merge_targets().astype(np.int64)
    src = np.nonzero(mt != np.arange(len(mt)))[0]
    assert len(src) > 1000                                                  # neighbours were merged
    assert np.all(rows["is_merged"][src] == 1) and np.all(rows["is_merged"][mt[src]] == 0)   # targets are final
    assert np.all(mt[mt[src]] == mt[src])                                   # no chains left (reassign, :64-82)
    whitelist_cells = set(int(x) for x in s.cell_cb)
    assert all(int(b) in whitelist_cells for b in rows["barcode"][mt[src]])   # every target is a whitelist barcode
    assert c.real_cells_number() < n_real_before
    cell, gene, umi, reads, mark = c.molecules()
    assert int(reads.sum()) + int(c.global_counters()[0]) == dev.n          # reads conserved through the unions
    assert not np.any(np.isin(cell, src))                                   # merged sources own no molecule any more
    # Stats::merge quirk: TOTAL_UMIS of a target = sum over its sources, >= its distinct molecule count
    real = rows["is_real"].astype(bool)
    per_cell = np.bincount(cell, minlength=len(rows))
    assert np.all(rows["total_umis"][real] >= per_cell[real])
    dev.free()


@pytest.mark.parametrize("shape", ["c3", "c5_share"])
def test_full_size_1e9_reads_properties(shape):
    """BASELINE configs[2] at FULL size (C3: 1e9 reads, 50 000 cells, UMI 12, -m + 10x whitelist) and the single-GPU share of
    configs[4] (C5: 1e9 of 8e9 reads, 62 500 of 500 000 cells, UMI 12, no CB merge): far beyond the oracle, so size-independent
    properties -- every read counted once, strictly ascending molecule keys, per-cell sums, CSC structure, merge targets final
    whitelist cells -- checked on the device-resident tables through chunked accessors."""
    c3 = shape == "c3"
    s = SynthStream(n_reads=1_000_000_000, n_cells=50_000 if c3 else 62_500, n_genes=30000, umi_len=12, stream_id=3 if c3 else 5)
    dev = s.generate_device(0)
    kw = dict(min_genes_before_merge=20, min_genes_after_merge=100)
    if c3:
        kw.update(merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=capi.BARCODES_CONST, barcodes_file=os.path.join(DATA, "10x_aug_2016_split"))
    c = capi.Context(**kw)
    c.push_reads_device(*dev.ptrs, dev.n, adopt=True)
    c.set_initialized()
    n_real_before = c.real_cells_number()
    c.merge_and_filter()
    L = c.sort_layout()
    assert L["cell_bits"] + L["gene_bits"] + L["umi_bits"] == 64 and L["value_bytes"] == 1 and L["sort"] == "splitter"
    sizes = c.table_sizes()
    assert sizes["reads"] == dev.n
    rows = c.cell_rows()
    counters = c.global_counters()
    real = rows["is_real"].astype(bool)
    merged = rows["is_merged"].astype(bool)
    # reads: every gene-bearing read sits in exactly one cell's TOTAL_READS; merged sources keep their stale counter while
    # the target's includes it (Stats::merge adds, Stats.cpp:29-43) -> count the cells that were never merged away
    own = rows["total_reads"].astype(np.int64)
    assert int(own[~merged].sum()) + int(counters[0]) == dev.n
    p, i, v = c.count_matrix_csc(filtered=False)
    assert len(p) - 1 == int(real.sum()) and np.all(np.diff(p.astype(np.int64)) == rows["n_genes"][real])
    assert np.all(i < 30000) and np.all(v > 0)
    # genes ascend strictly inside every column
    d = np.diff(i.astype(np.int64)); starts = p[1:-1].astype(np.int64)
    d[starts - 1] = 1
    assert np.all(d > 0)
    p2, i2, v2 = c.count_matrix_csc(filtered=True)
    f = c.filtered_cells().astype(np.int64)
    key = np.stack([rows["requested_genes"][f], rows["requested_umis"][f], rows["total_umis"][f]], axis=1).astype(np.int64)
    assert np.all(np.lexsort((key[:, 2], key[:, 1], key[:, 0])) == np.arange(len(f))) or np.all(np.diff(key[:, 0]) >= 0)
    assert int(key[:, 0].min()) >= 100 and np.all(np.diff(p2.astype(np.int64)) == rows["requested_genes"][f])
    assert int(v2.astype(np.int64).sum()) == int(rows["requested_umis"][f].astype(np.int64).sum())
    if c3:
        mt = c.merge_targets().astype(np.int64)
        src = np.nonzero(mt != np.arange(len(mt)))[0]
        assert len(src) > 10_000 and np.all(rows["is_merged"][src] == 1) and np.all(mt[mt[src]] == mt[src])
        wl_cells = set(int(x) for x in s.cell_cb)
        assert all(int(b) in wl_cells for b in rows["barcode"][mt[src[:20000]]])
        assert c.real_cells_number() < n_real_before
    else:
        assert int(real.sum()) >= 62_500 and int(rows["total_umis"].astype(np.int64)[real].sum()) == int(v.astype(np.int64).sum())
    # a slice of the molecule table: strictly ascending (gene, umi) inside a cell, reads add up to the cell's counter
    big = int(np.argmax(rows["total_reads"] * real))
    g, u, r, m = c.cell_molecules(big)
    k = (g.astype(np.uint64) << np.uint64(32)) | (u & np.uint64((1 << 24) - 1))
    assert np.all(k[1:] > k[:-1]) and len(np.unique(g)) == rows["n_genes"][big]
    del p, i, v, p2, i2, v2, d
    _wire_equals_direct(c)
    dev.free()


# ---------------------------------------------------------------------------------------------------
# CB merge against a whitelist (RealBarcodesMergeStrategy)
# ---------------------------------------------------------------------------------------------------
import os

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dropest_amd", "data", "barcodes")

FIXTURE_READS = [   # Tests/TestEstimation.cpp:49-74 (struct Fixture)
    ("AAATTAGGTCCA", "AAACCT", "Gene1"), ("AAATTAGGTCCA", "CCCCCT", "Gene2"), ("AAATTAGGTCCA", "ACCCCT", "Gene3"),
    ("AAATTAGGTCCA", "ACCCCT", "Gene4"), ("AAATTAGGTCCC", "CAACCT", "Gene1"), ("AAATTAGGTCCC", "CAACCT", "Gene10"),
    ("AAATTAGGTCCC", "CAACCT", "Gene20"), ("AAATTAGGTCCG", "CAACCT", "Gene1"), ("AAATTAGGTCGG", "AAACCT", "Gene1"),
    ("AAATTAGGTCGG", "CCCCCT", "Gene2"), ("CCCTTAGGTCCA", "CCATTC", "Gene3"), ("CCCTTAGGTCCA", "CCCCCT", "Gene2"),
    ("CCCTTAGGTCCA", "ACCCCT", "Gene3"), ("CAATTAGGTCCG", "CAACCT", "Gene1"), ("CAATTAGGTCCG", "AAACCT", "Gene1"),
    ("CAATTAGGTCCG", "CCCCCT", "Gene2"), ("AAAAAAAAAAAA", "CCCCCT", "Gene2"),
]


def _pack_reads(reads):
    genes = {}
    cb = np.array([capi.pack_seq(r[0]) for r in reads], np.uint64)
    umi = np.array([capi.pack_seq(r[1]) for r in reads], np.uint64)
    gene = np.array([genes.setdefault(r[2], len(genes)) for r in reads], np.uint32)
    aux = np.full(len(reads), 2 << 16, np.uint32)
    return cb, umi, gene, aux, list(genes)


def test_reference_fixture_merge_by_real_barcodes():
    """Tests/TestEstimation.cpp:227-280 (testRealNeighbours, testMergeByRealBarcodes) through the C-ABI on the GPU."""
    cb, umi, gene, aux, names = _pack_reads(FIXTURE_READS)
    kw = dict(merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=capi.BARCODES_INDROP,
              barcodes_file=os.path.join(DATA, "test_est"), min_genes_before_merge=0, min_genes_after_merge=0,
              max_cb_merge_edit_distance=7, min_merge_fraction=0.0)
    c = capi.Context(**kw)
    c.push_reads(cb, umi, gene, aux)
    c.set_initialized()
    assert [c.merge_target(i) for i in range(6)] == [0, 1, 1, 0, 0, 0]          # :227-235
    assert c.merge_target(6) == -1
    c.merge_and_filter()
    assert c.total_cells_number() == 7
    f = c.filtered_cells()
    assert len(f) == 2
    rows = c.cell_rows()
    assert rows["n_genes"][int(f[0])] == 3 and rows["n_genes"][int(f[1])] == 4
    assert list(rows["is_merged"]) == [0, 0, 1, 1, 1, 1, 0]
    assert int(rows["is_excluded"].sum()) == 1
    assert list(c.merge_targets()) == [0, 1, 1, 0, 0, 0, 6]

    def mol(cell):
        g, u, r, m = c.cell_molecules(cell)
        out = {}
        for gi, ui, ri in zip(g, u, r):
            out.setdefault(names[int(gi)], {})[capi.unpack_code(ui)] = int(ri)
        return out
    m0, m1 = mol(int(f[0])), mol(int(f[1]))
    assert m0["Gene1"] == {"CAACCT": 2}
    assert m1["Gene1"] == {"AAACCT": 3, "CAACCT": 1} and m1["Gene2"] == {"CCCCCT": 4}
    assert m1["Gene3"] == {"ACCCCT": 2, "CCATTC": 1}
    # and everything else against the oracle
    o = parity.oracle_run(Oracle, dict(merge_kind=1, barcodes_kind=0, barcodes_file=os.path.join(DATA, "test_est"),
                                       min_genes_before=0, min_genes_after=0, max_cb_merge_ed=7, min_merge_fraction=0.0),
                          cb, umi, gene, aux)
    parity.compare(o, c)


def _both_merge(stream_kw, n_reads, min_before, min_after, whitelist, kind, frac=0.2):
    s = SynthStream(n_reads=n_reads, whitelist=whitelist, **stream_kw)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    path = os.path.join(DATA, whitelist)
    o = parity.oracle_run(Oracle, dict(merge_kind=1, barcodes_kind=kind, barcodes_file=path, min_genes_before=min_before,
                                       min_genes_after=min_after, min_merge_fraction=frac), cb, umi, gene, aux)
    c = parity.gpu_run(dict(merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=kind, barcodes_file=path,
                            min_genes_before_merge=min_before, min_genes_after_merge=min_after,
                            min_merge_fraction=frac), cb, umi, gene, aux)
    parity.compare(o, c)
    return o, c


def test_c3_shape_merge_10x_whitelist():
    """10x v3 shape with Hamming-1 neighbour barcodes and -m + whitelist (C3 scaled down)."""
    o, c = _both_merge(dict(n_cells=30, n_genes=2000, umi_len=12, permille_neighbour=150), 200_000, 3, 20,
                       "10x_aug_2016_split", capi.BARCODES_CONST)
    mt = c.merge_targets()
    assert int((mt != np.arange(len(mt))).sum()) > 50          # neighbours really merged
    assert int(c.cell_rows()["is_excluded"].sum()) > 0         # and unrelated barcodes excluded


def test_c4_shape_merge_indrop_v3_whitelist():
    """inDrop v3 shape: split 8+8 barcode, UMI 8, const-length parser (C4 scaled down)."""
    o, c = _both_merge(dict(n_cells=25, n_genes=1500, umi_len=8, permille_neighbour=120), 120_000, 3, 10,
                       "indrop_v3", capi.BARCODES_CONST)
    mt = c.merge_targets()
    assert int((mt != np.arange(len(mt))).sum()) > 20


def test_merge_zero_fraction_threshold_order_dependence():
    """min_merge_fraction = 0: candidates with an empty intersection still win (first in the reference's order)."""
    _both_merge(dict(n_cells=20, n_genes=800, umi_len=10, permille_neighbour=200), 60_000, 2, 5,
                "10x_aug_2016_split", capi.BARCODES_CONST, frac=0.0)


# ---------------------------------------------------------------------------------------------------
# UMIs containing N (MergeUMIsStrategySimple)
# ---------------------------------------------------------------------------------------------------
from dropest_amd.synth import inject_n


def _pack_with_escapes(reads):
    genes, side, index = {}, [], {}
    def code(s):
        c = capi.pack_seq(s)
        if c is not None:
            return c
        k = index.get(s)
        if k is None:
            k = index[s] = len(side)
            side.append(s)
        return capi.ESCAPE | k
    cb = np.array([code(r[0]) for r in reads], np.uint64)
    umi = np.array([code(r[1]) for r in reads], np.uint64)
    gene = np.array([genes.setdefault(r[2], len(genes)) for r in reads], np.uint32)
    aux = np.full(len(reads), 2 << 16, np.uint32)
    return cb, umi, gene, aux, list(genes), side


def test_reference_fixture_umi_merge_strategy_simple():
    """Tests/TestEstimation.cpp:505-540 testUMIMergeStrategySimple through the C-ABI."""
    cbs = "AAATTAGGTCCA"
    reads = [(cbs, u, "Gene1") for u in ["AAACCT", "AAACCT", "AAACCG", "AAACCN", "CCCCCT", "ACCCCT"]]
    reads += [(cbs, u, "Gene2") for u in ["TTTTTT", "TTTNNG", "TTGNNG", "ACCCCT", "NNNNNN"]]
    cb, umi, gene, aux, names, side = _pack_with_escapes(reads)
    c = capi.Context(min_genes_before_merge=0, min_genes_after_merge=0)
    c.set_side_strings(side)
    c.push_reads(cb, umi, gene, aux)
    c.set_initialized(); c.merge_and_filter()
    g, u, r, m = c.cell_molecules(0)
    mol = {}
    for gi, ui, ri in zip(g, u, r):
        mol.setdefault(names[int(gi)], {})[capi.unpack_code(ui, side)] = int(ri)
    assert len(mol["Gene1"]) == 4 and len(mol["Gene2"]) == 3
    assert mol["Gene1"] == {"AAACCT": 3, "AAACCG": 1, "CCCCCT": 1, "ACCCCT": 1}
    assert "TTTTTT" in mol["Gene2"] and "ACCCCT" in mol["Gene2"]
    assert all("N" not in x for x in mol["Gene2"])
    o = parity.oracle_run(Oracle, dict(min_genes_before=0, min_genes_after=0), cb, umi, gene, aux, side)
    parity.compare(o, c, side)          # includes the glibc rand() fills and the TOTAL_UMIS decrements


def test_survey_probe_random_fill():
    """SURVEY.md §7 probe: gene {ACGTCC, NNNNAA} ends as {ACGTCC, GACCAA}, umis_number() == 1."""
    reads = [("AAAA", "ACGTCC", "G"), ("AAAA", "NNNNAA", "G")]
    cb, umi, gene, aux, names, side = _pack_with_escapes(reads)
    c = capi.Context(min_genes_before_merge=0, min_genes_after_merge=0)
    c.set_side_strings(side); c.push_reads(cb, umi, gene, aux)
    c.set_initialized(); c.merge_and_filter()
    g, u, r, m = c.cell_molecules(0)
    assert sorted(capi.unpack_code(x, side) for x in u) == ["ACGTCC", "GACCAA"]
    assert c.cell_rows()["total_umis"][0] == 1


@pytest.mark.parametrize("rate,n_reads", [(1e-2, 120_000), (1e-3, 600_000)])
def test_n_umis_synthetic(rate, n_reads):
    s = SynthStream(n_reads=n_reads, n_cells=40, n_genes=1500, umi_len=8)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    umi, side = inject_n(umi, gene, rate, 7, 8)
    assert len(side) > 20
    o = parity.oracle_run(Oracle, dict(min_genes_before=10, min_genes_after=20), cb, umi, gene, aux, side)
    c = parity.gpu_run(dict(min_genes_before_merge=10, min_genes_after_merge=20), cb, umi, gene, aux, side)
    parity.compare(o, c, side)


def test_n_umis_with_cb_merge():
    s = SynthStream(n_reads=150_000, n_cells=25, n_genes=1200, umi_len=8, permille_neighbour=150)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    umi, side = inject_n(umi, gene, 5e-3, 11, 8)
    path = os.path.join(DATA, "10x_aug_2016_split")
    o = parity.oracle_run(Oracle, dict(merge_kind=1, barcodes_kind=1, barcodes_file=path, min_genes_before=3,
                                       min_genes_after=10), cb, umi, gene, aux, side)
    c = parity.gpu_run(dict(merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=capi.BARCODES_CONST, barcodes_file=path,
                            min_genes_before_merge=3, min_genes_after_merge=10), cb, umi, gene, aux, side)
    parity.compare(o, c, side)


def test_device_ordering_of_filtered_cells(monkeypatch):
    """Large filtered lists are ordered by three stable radix sorts on the device; force that path on a stream
    small enough for the oracle (DROPEST_DEVICE_SORT_MIN) -- with and without the whitelist merge."""
    monkeypatch.setenv("DROPEST_DEVICE_SORT_MIN", "1")
    _both(dict(n_cells=60, n_genes=3000), 150_000, 2, 5)
    _both_merge(dict(n_cells=30, n_genes=2000, umi_len=12, permille_neighbour=150), 200_000, 3, 20,
                "10x_aug_2016_split", capi.BARCODES_CONST)


# ---------------------------------------------------------------------------------------------------
# sort-record layouts: keys only / key + 1-byte mark / key + (chromosome | mark)
# ---------------------------------------------------------------------------------------------------
LAYOUTS = {"keys": (None, "rs_scatter:keys"), "byte": ("DROPEST_FORCE_BYTE_VALUES", "rs_scatter:key+1B"),
           "general": ("DROPEST_FORCE_GENERAL_LAYOUT", "rs_scatter")}


@pytest.mark.parametrize("layout", sorted(LAYOUTS))
@pytest.mark.parametrize("case", ["plain", "n_umis", "cb_merge"])
def test_sort_layouts_agree_with_oracle(monkeypatch, layout, case):
    """The three record layouts must give identical results on a stream whose genes each sit on one chromosome
    (the synthetic generator's); which one ran is read back from the kernel statistics."""
    env, kernel = LAYOUTS[layout]
    if env:
        monkeypatch.setenv(env, "1")
    okw = dict(min_genes_before=3, min_genes_after=10)
    gkw = dict(min_genes_before_merge=3, min_genes_after_merge=10)
    side = ()
    if case == "cb_merge":
        s = SynthStream(n_reads=150_000, n_cells=25, n_genes=1200, umi_len=8, permille_neighbour=150,
                        whitelist="10x_aug_2016_split")
        path = os.path.join(DATA, "10x_aug_2016_split")
        okw.update(merge_kind=1, barcodes_kind=1, barcodes_file=path)
        gkw.update(merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=capi.BARCODES_CONST, barcodes_file=path)
    else:
        s = SynthStream(n_reads=120_000, n_cells=40, n_genes=1500, umi_len=8, permille_intergenic=120)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    if case != "plain":
        umi, side = inject_n(umi, gene, 5e-3, 5, 8)
    o = parity.oracle_run(Oracle, okw, cb, umi, gene, aux, side)
    c = parity.gpu_run(gkw, cb, umi, gene, aux, side, chunks=2, profile=True)
    parity.compare(o, c, side)
    names = set(c.kernel_stats())
    assert kernel in names and not (names & {k for _, k in LAYOUTS.values()} - {kernel, "rs_scatter"}), names
    assert ("seg_reduce:chr_rows" in names) == (layout == "general")


def test_gene_on_two_chromosomes_falls_back_to_general_layout():
    """One gene reported on two chromosomes: chromosome counters can no longer be derived from (cell, gene) rows."""
    s = SynthStream(n_reads=60_000, n_cells=20, n_genes=500, umi_len=8)
    cb, umi, gene, aux = s.generate_host()
    hit = np.flatnonzero((gene != capi.NO_GENE) & (((aux >> 16) & 6) != 0))
    i = int(hit[-1]); old = int(aux[i]) & 0xFFFF
    aux = aux.copy(); aux[i] = (int(aux[i]) & 0xFFFF0000) | ((old + 1) % 3)          # always != old
    cb, umi, gene, aux = parity.canonical_stream(cb, umi, gene, aux)
    o = parity.oracle_run(Oracle, dict(min_genes_before=3, min_genes_after=10), cb, umi, gene, aux)
    c = parity.gpu_run(dict(min_genes_before_merge=3, min_genes_after_merge=10), cb, umi, gene, aux, profile=True)
    parity.compare(o, c)
    assert "seg_reduce:chr_rows" in c.kernel_stats()


# ---------------------------------------------------------------------------------------------------
# UMI distribution + Tools::CollisionsAdjuster
# ---------------------------------------------------------------------------------------------------
from oracle import binding as ob


def test_umi_distribution_matches_oracle():
    """CellsDataContainer::umi_distribution (CellsDataContainer.cpp:182-197), also after the N-UMI merge."""
    s = SynthStream(n_reads=150_000, n_cells=40, n_genes=1500, umi_len=6)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    umi, side = inject_n(umi, gene, 5e-3, 3, 6)
    o = parity.oracle_run(Oracle, dict(min_genes_before=10, min_genes_after=20), cb, umi, gene, aux, side)
    c = parity.gpu_run(dict(min_genes_before_merge=10, min_genes_after_merge=20), cb, umi, gene, aux, side)
    codes, counts = c.umi_distribution()
    names, ocounts = o.umi_distribution()
    got = {capi.unpack_code(u, side): int(n) for u, n in zip(codes, counts)}
    assert got == dict(zip(names, [int(x) for x in ocounts]))
    assert len(got) > 1000 and all("N" not in k for k in got)
    assert np.all(np.diff(codes.astype(np.float64)) > 0)            # ascending UMI code


def test_collisions_adjuster_table():
    """Tools::CollisionsAdjuster (Tools/CollisionsAdjuster.cpp:12-49).  The recurrence is evaluated in double on the
    device with a fixed-order parallel sum; the oracle sums left to right like the reference.  Bar (stated in
    include/dropest_amd.h): the integer table is identical; the reference itself pins this component to 1e-2 only."""
    rng = np.random.default_rng(12)
    # (a) the UMI distribution of a synthetic run, normalised as PoissonTargetEstimator::init does (:46-60)
    s = SynthStream(n_reads=200_000, n_cells=40, n_genes=1500, umi_len=6)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    c = parity.gpu_run(dict(min_genes_before_merge=10, min_genes_after_merge=20), cb, umi, gene, aux)
    _, counts = c.umi_distribution()
    p = counts.astype(np.float64) / float(counts.sum())
    smax = len(p) // 3          # the recurrence diverges once the expression approaches the size of the UMI space
    assert np.array_equal(capi.collisions_adjusted_sizes(p, smax), ob.collisions_table(p, smax))
    # (b) uniform and skewed toy distributions (expression well below the UMI-space size, as in real use)
    for n, skew, smax in ((16, 0.0, 8), (64, 1.0, 20), (4096, 0.5, 1500), (4096, 0.0, 3000)):
        w = 1.0 / np.arange(1, n + 1) ** skew
        w = rng.permutation(w / w.sum())
        got, want = capi.collisions_adjusted_sizes(w, smax), ob.collisions_table(w, smax)
        assert np.array_equal(got, want), (n, skew)
        assert np.all(np.diff(got.astype(np.int64)) >= 1)                   # adjusted sizes grow with s


# ---------------------------------------------------------------------------------------------------
# -u: MergeUMIsStrategyDirectional (Estimation/Merge/UMIs/MergeUMIsStrategyDirectional.cpp:18-116)
# ---------------------------------------------------------------------------------------------------
import ctypes as _C

_libc = _C.CDLL("libc.so.6")


def _both_directional(cb, umi, gene, aux, side=(), mult=2.0, max_ed=1, extra_o=None, extra_g=None, min_genes=1):
    """The directional strategy never seeds rand(): both runs start from the same explicit state (a fresh reference
    process starts from srand(1))."""
    okw = dict(umi_merge_kind=1, max_umi_merge_ed=max_ed, umi_mult=mult, min_genes_before=min_genes, min_genes_after=min_genes)
    gkw = dict(umi_merge_kind=capi.UMI_MERGE_DIRECTIONAL, max_umi_merge_edit_distance=max_ed, umi_merge_multiplier=mult,
               min_genes_before_merge=min_genes, min_genes_after_merge=min_genes)
    okw.update(extra_o or {}); gkw.update(extra_g or {})
    _libc.srand(1)
    o = parity.oracle_run(Oracle, okw, cb, umi, gene, aux, side)
    _libc.srand(1)
    c = parity.gpu_run(gkw, cb, umi, gene, aux, side, profile=True)
    parity.compare(o, c, side)
    return o, c


def test_directional_reference_fixture_on_gpu():
    """The UMI list of testUMIMergeStrategyDirectional (Tests/TestEstimation.cpp:588-608) as reads of one (cell, gene):
    AAA and AAT collapse into AGT, CCC into TCC."""
    umis = [("AAA", 2), ("AAC", 5), ("AAT", 6), ("AGT", 20), ("CCC", 10), ("TCC", 20)]
    seqs = [u for u, n in umis for _ in range(n)]
    # first occurrences in the listed order, the remaining reads shuffled
    rng = np.random.default_rng(3)
    rest = [u for u, n in umis for _ in range(n - 1)]
    rng.shuffle(rest)
    seqs = [u for u, _ in umis] + rest
    n = len(seqs)
    cb = np.full(n, capi.pack_seq("ACGTACGTACGT"), np.uint64)
    umi = np.array([capi.pack_seq(s) for s in seqs], np.uint64)
    gene = np.zeros(n, np.uint32); aux = np.full(n, 2 << 16, np.uint32)
    o, c = _both_directional(cb, umi, gene, aux)
    _, g, u, r, m = c.molecules()
    got = {capi.unpack_code(x): int(y) for x, y in zip(u, r)}
    assert got == {"AGT": 28, "AAC": 5, "TCC": 30}
    assert int(c.cell_rows()["total_umis"][0]) == 6 - 3                 # one decrement per re-keyed UMI


@pytest.mark.parametrize("umi_len,n_genes,max_ed,mult", [(5, 300, 1, 2.0), (6, 150, 2, 1.5), (4, 60, 1, 1.0), (8, 400, 3, 2.0)])
def test_directional_synthetic(umi_len, n_genes, max_ed, mult):
    """Short UMIs on few genes: many Hamming-1 neighbours inside a (cell, gene) group, groups of 2..16 UMIs decided on
    the device and larger ones on the host."""
    s = SynthStream(n_reads=150_000, n_cells=30, n_genes=n_genes, umi_len=umi_len)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    o, c = _both_directional(cb, umi, gene, aux, mult=mult, max_ed=max_ed, min_genes=5)
    st = c.kernel_stats()
    assert "umi_directional" in st and ("fold:cell_gene" in st or "seg_reduce:molecules_rekeyed" in st)   # the device path re-keyed molecules
    has_gene = gene != capi.NO_GENE
    distinct = np.unique(np.stack([cb[has_gene], gene[has_gene].astype(np.uint64), umi[has_gene]]), axis=1).shape[1]
    assert int(c.molecules()[0].shape[0]) < distinct * 0.98                       # UMIs really collapsed


def test_directional_with_n_umis_and_cb_merge():
    """N-UMIs (random fills from rand()) and the whitelist CB merge before the UMI correction."""
    s = SynthStream(n_reads=150_000, n_cells=25, n_genes=400, umi_len=6, permille_neighbour=150)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    umi, side = inject_n(umi, gene, 1e-2, 5, 6)
    path = os.path.join(DATA, "10x_aug_2016_split")
    o, c = _both_directional(cb, umi, gene, aux, side, min_genes=3,
                             extra_o=dict(merge_kind=1, barcodes_kind=1, barcodes_file=path),
                             extra_g=dict(merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=capi.BARCODES_CONST, barcodes_file=path))
    assert int((c.merge_targets() != np.arange(c.total_cells_number())).sum()) > 20
    codes, counts = c.umi_distribution()
    assert all("N" not in capi.unpack_code(u, side) for u in codes)


def test_directional_all_layouts(monkeypatch):
    for env in (None, "DROPEST_FORCE_BYTE_VALUES", "DROPEST_FORCE_GENERAL_LAYOUT"):
        if env:
            monkeypatch.setenv(env, "1")
        s = SynthStream(n_reads=80_000, n_cells=20, n_genes=200, umi_len=5, permille_intergenic=100)
        cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
        _both_directional(cb, umi, gene, aux, min_genes=5)
        if env:
            monkeypatch.delenv(env)


def test_directional_very_large_groups():
    """(cell, gene) groups beyond the LDS kernel's 4096 UMIs (one hot gene in one cell) run in the global-scratch kernel:
    introsort on 5 000+ elements with heavy ties in the read counts, against the oracle's real std::sort."""
    rng = np.random.default_rng(5)
    n_umis = [6000, 4097, 4096, 900]                                   # per (cell, gene) group
    cb, umi, gene = [], [], []
    for g, k in enumerate(n_umis):
        codes = rng.choice(4 ** 7, size=k, replace=False).astype(np.uint64) | np.uint64(1 << 14)
        reps = rng.choice([1, 1, 1, 2, 2, 3, 5, 9], size=k)
        u = np.repeat(codes, reps)
        rng.shuffle(u)
        umi.append(u); gene.append(np.full(len(u), g % 2, np.uint32)); cb.append(np.full(len(u), capi.pack_seq("ACGTACGTAC" + "AC"[g // 2] * 2), np.uint64))
    cb, umi, gene = np.concatenate(cb), np.concatenate(umi), np.concatenate(gene)
    perm = rng.permutation(len(cb))
    cb, umi, gene = cb[perm], umi[perm], gene[perm]
    aux = np.full(len(cb), 2 << 16, np.uint32)
    cb, umi, gene, aux = parity.canonical_stream(cb, umi, gene, aux)
    o, c = _both_directional(cb, umi, gene, aux, min_genes=1)
    st = c.kernel_stats()
    assert st["count:umi_groups_huge"]["launches"] == 2 and st["count:umi_groups_wave"]["launches"] == 2
    assert st["count:umi_groups_host"]["launches"] == 0 and st["count:umi_rekeyed"]["launches"] > 1000


# ---------------------------------------------------------------------------------------------------
# -m without a whitelist: SimpleMergeStrategy (Estimation/Merge/SimpleMergeStrategy.cpp)
# ---------------------------------------------------------------------------------------------------
def _both_simple(cb, umi, gene, aux, side=(), max_ed=2, frac=0.2, min_before=3, min_after=10):
    o = parity.oracle_run(Oracle, dict(merge_kind=2, max_cb_merge_ed=max_ed, min_merge_fraction=frac, min_genes_before=min_before,
                                       min_genes_after=min_after), cb, umi, gene, aux, side)
    c = parity.gpu_run(dict(merge_kind=capi.MERGE_SIMPLE, max_cb_merge_edit_distance=max_ed, min_merge_fraction=frac,
                            min_genes_before_merge=min_before, min_genes_after_merge=min_after), cb, umi, gene, aux, side, profile=True)
    parity.compare(o, c, side)
    return o, c


@pytest.mark.parametrize("max_ed,frac,umi_len", [(2, 0.2, 8), (3, 0.05, 8), (7, 0.0, 6), (2, 0.2, 5)])
def test_simple_merge_synthetic(max_ed, frac, umi_len):
    """Hamming-1 error barcodes share their cell's molecules: they merge into it when no whitelist is given.  Short
    UMIs add chance collisions between unrelated cells (near-tie replays); frac = 0 keeps every candidate alive."""
    s = SynthStream(n_reads=150_000, n_cells=25, n_genes=1200, umi_len=umi_len, permille_neighbour=150)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    o, c = _both_simple(cb, umi, gene, aux, max_ed=max_ed, frac=frac)
    mt = c.merge_targets()
    assert int((mt != np.arange(len(mt))).sum()) > 300
    assert int(c.cell_rows()["is_excluded"].sum()) == 0                    # this strategy never excludes


def test_simple_merge_ties_are_replayed():
    """Few distinct UMIs and genes: many candidates with exactly equal fractions and sizes, so the reference's answer
    hangs on the iteration order of its unordered containers -- the replay path must reproduce it."""
    s = SynthStream(n_reads=60_000, n_cells=12, n_genes=40, umi_len=3, permille_neighbour=250)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    o, c = _both_simple(cb, umi, gene, aux, max_ed=17, frac=0.0, min_before=1, min_after=1)
    assert "host:cb_merge:replay" in c.kernel_stats()


def test_simple_merge_with_n_and_directional():
    s = SynthStream(n_reads=100_000, n_cells=20, n_genes=600, umi_len=6, permille_neighbour=150)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    umi, side = inject_n(umi, gene, 5e-3, 9, 6)
    _libc.srand(1)
    o = parity.oracle_run(Oracle, dict(merge_kind=2, max_cb_merge_ed=2, umi_merge_kind=1, min_genes_before=3, min_genes_after=10),
                          cb, umi, gene, aux, side)
    _libc.srand(1)
    c = parity.gpu_run(dict(merge_kind=capi.MERGE_SIMPLE, max_cb_merge_edit_distance=2, umi_merge_kind=capi.UMI_MERGE_DIRECTIONAL,
                            min_genes_before_merge=3, min_genes_after_merge=10), cb, umi, gene, aux, side)
    parity.compare(o, c, side)


# ---------------------------------------------------------------------------------------------------
# -M with a whitelist: PoissonRealBarcodesMergeStrategy + PoissonTargetEstimator
# ---------------------------------------------------------------------------------------------------
POISSON_FIXTURE = [("AAATTAGGTCCA", "AAACCT", "Gene1"), ("AAATTAGGTCCA", "CCCCCT", "Gene2"), ("AAATTAGGTCCA", "ACCCCT", "Gene3"),
                   ("AAATTAGGTCCC", "CAACCT", "Gene1"), ("AAATTAGGTCCG", "CAACCT", "Gene1"),
                   ("AAATTAGGTCGG", "AAACCT", "Gene1"), ("AAATTAGGTCGG", "CCCCCT", "Gene2"),
                   ("CCCTTAGGTCCA", "CCATTC", "Gene3"), ("CCCTTAGGTCCA", "CCCCCT", "Gene2"), ("CCCTTAGGTCCA", "ACCCCT", "Gene3"),
                   ("CAATTAGGTCCG", "CAACCT", "Gene1"), ("CAATTAGGTCCG", "AAACCT", "Gene1"), ("CAATTAGGTCCG", "CCCCCT", "Gene2"),
                   ("CAATTAGGTCCG", "TTTTTT", "Gene2"), ("CAATTAGGTCCG", "TTCTTT", "Gene2"),
                   ("CCCCCCCCCCCC", "CAACCT", "Gene1"), ("CCCCCCCCCCCC", "AAACCT", "Gene1"), ("CCCCCCCCCCCC", "CCCCCT", "Gene2"),
                   ("CCCCCCCCCCCC", "TTTTTT", "Gene2"), ("CCCCCCCCCCCC", "TTCTTT", "Gene2"), ("TAATTAGGTCCA", "AAAAAA", "Gene4")]


def _poisson_kw(path, kind, min_before, min_after, p_merge=1e-4, p_real=1e-7):
    okw = dict(merge_kind=3, barcodes_kind=kind, barcodes_file=path, min_genes_before=min_before, min_genes_after=min_after,
               max_merge_prob=p_merge, max_real_merge_prob=p_real)
    gkw = dict(merge_kind=capi.MERGE_POISSON_REAL, barcodes_kind=kind, barcodes_file=path, min_genes_before_merge=min_before,
               min_genes_after_merge=min_after, max_merge_prob=p_merge, max_real_merge_prob=p_real)
    return okw, gkw


def test_poisson_reference_fixture_on_gpu():
    """Tests/TestEstimationMergeProbs.cpp:30-140 through the C-ABI: testPoissonMergeProbs' tolerances (three of the four
    values; see tests/test_oracle_reference_kat.py for the fourth), testPoissonMergeRejections, and the estimator's
    numbers against the oracle's for every cell pair."""
    cb, umi, gene, aux, names = _pack_reads(POISSON_FIXTURE)
    okw, gkw = _poisson_kw(os.path.join(DATA, "test_est"), 0, 0, 0)
    c = capi.Context(**gkw)
    c.push_reads(cb, umi, gene, aux)
    c.set_initialized()
    o = Oracle(**okw)
    o.add_packed(cb, umi, gene, aux, ())
    o.set_initialized()
    o.poisson_init()
    assert c.poisson_intersection_prob(0, 1)[2] == 1                                   # :129
    assert abs(c.poisson_intersection_prob(1, 2)[2] - 0.16) <= 0.05                    # :130
    assert abs(c.poisson_intersection_prob(3, 4)[2] - 0.15) <= 0.05                    # :131
    n = c.total_cells_number()
    for i in range(n):
        for j in range(n):
            if i == j:
                continue
            inter, expected, prob = c.poisson_intersection_prob(i, j)
            want_e, want_p = o.poisson_expected_intersection(i, j), o.poisson_intersection_prob(i, j)
            if inter == 0:
                assert expected == -1 and prob == 1 and want_p == 1
            else:
                assert abs(expected - want_e) <= 1e-12 * max(1.0, want_e), (i, j, expected, want_e)
                assert abs(prob - want_p) <= 1e-10 * want_p + 1e-300, (i, j, prob, want_p)
    assert c.merge_target(7) == -1 == o.poisson_merge_target(7)                        # :136-140
    assert [c.merge_target(i) for i in range(n)] == [o.poisson_merge_target(i) for i in range(n)]
    c.merge_and_filter()
    o.merge_and_filter()
    parity.compare(o, c)


def _both_poisson(stream_kw, n_reads, min_before, min_after, whitelist, kind, **probs):
    s = SynthStream(n_reads=n_reads, whitelist=whitelist, **stream_kw)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    okw, gkw = _poisson_kw(os.path.join(DATA, whitelist), kind, min_before, min_after, **probs)
    o = parity.oracle_run(Oracle, okw, cb, umi, gene, aux)
    c = parity.gpu_run(gkw, cb, umi, gene, aux)
    parity.compare(o, c)
    return o, c


def test_poisson_merge_10x_whitelist():
    o, c = _both_poisson(dict(n_cells=30, n_genes=2000, umi_len=12, permille_neighbour=150), 200_000, 3, 20,
                         "10x_aug_2016_split", capi.BARCODES_CONST)
    mt = c.merge_targets()
    assert int((mt != np.arange(len(mt))).sum()) >= 12
    assert int(c.cell_rows()["is_excluded"].sum()) > 0


def test_poisson_merge_indrop_v3_whitelist_and_loose_thresholds():
    """Short UMIs (8 bases: collisions are frequent, the adjuster matters) and thresholds loose enough that real
    barcodes merge into each other (PoissonTargetEstimator.cpp:17-20 uses max_merge_prob for REAL bases)."""
    _both_poisson(dict(n_cells=25, n_genes=1500, umi_len=8, permille_neighbour=120), 120_000, 3, 10, "indrop_v3", capi.BARCODES_CONST)
    _both_poisson(dict(n_cells=25, n_genes=300, umi_len=8, permille_neighbour=120), 120_000, 3, 10, "indrop_v3", capi.BARCODES_CONST,
                  p_merge=0.5, p_real=0.5)


def test_poisson_probabilities_against_oracle_on_synthetic_pairs():
    s = SynthStream(n_reads=80_000, whitelist="10x_aug_2016_split", n_cells=20, n_genes=60, umi_len=6, permille_neighbour=150)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    okw, gkw = _poisson_kw(os.path.join(DATA, "10x_aug_2016_split"), capi.BARCODES_CONST, 3, 10)
    c = capi.Context(**gkw)
    c.push_reads(cb, umi, gene, aux)
    c.set_initialized()
    o = Oracle(**okw)
    o.add_packed(cb, umi, gene, aux, ())
    o.set_initialized()
    o.poisson_init()
    f = np.array([int(x) for x in c.filtered_cells()])
    big = f[np.argsort(-c.cell_rows()["total_umis"][f].astype(np.int64), kind="stable")[:9]]   # the cells that do intersect
    checked = 0
    for i, j in ((int(a), int(b)) for a in big for b in big if a != b):
        inter, expected, prob = c.poisson_intersection_prob(i, j)
        want_p = o.poisson_intersection_prob(i, j)
        if inter == 0:
            assert prob == 1 == want_p
            continue
        want_e = o.poisson_expected_intersection(i, j)
        assert abs(expected - want_e) <= 1e-11 * want_e, (i, j, expected, want_e)
        assert abs(prob - want_p) <= 1e-9 * want_p + 1e-300, (i, j, prob, want_p)
        checked += 1
    assert checked > 10


# ---------------------------------------------------------------------------------------------------
# -M without a whitelist: PoissonSimpleMergeStrategy
# ---------------------------------------------------------------------------------------------------
def _both_poisson_simple(cb, umi, gene, aux, side=(), max_ed=2, p_real=1e-7, min_before=3, min_after=10):
    o = parity.oracle_run(Oracle, dict(merge_kind=4, max_cb_merge_ed=max_ed, max_real_merge_prob=p_real, min_genes_before=min_before,
                                       min_genes_after=min_after), cb, umi, gene, aux, side)
    c = parity.gpu_run(dict(merge_kind=capi.MERGE_POISSON_SIMPLE, max_cb_merge_edit_distance=max_ed, max_real_merge_prob=p_real,
                            min_genes_before_merge=min_before, min_genes_after_merge=min_after), cb, umi, gene, aux, side)
    parity.compare(o, c, side)
    return o, c


@pytest.mark.parametrize("max_ed,p_real,umi_len", [(2, 1e-7, 10), (1, 1e-3, 8), (3, 0.5, 8)])
def test_poisson_simple_merge_synthetic(max_ed, p_real, umi_len):
    s = SynthStream(n_reads=150_000, whitelist="10x_aug_2016_split", n_cells=30, n_genes=1500, umi_len=umi_len, permille_neighbour=150)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    o, c = _both_poisson_simple(cb, umi, gene, aux, max_ed=max_ed, p_real=p_real)
    mt = c.merge_targets()
    assert int((mt != np.arange(len(mt))).sum()) > 20


# ---------------------------------------------------------------------------------------------------
# merge_type = "all": MergeAllMergeStrategy
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("max_ed", [1, 2, 4])
def test_merge_all_synthetic(max_ed):
    s = SynthStream(n_reads=150_000, whitelist="10x_aug_2016_split", n_cells=30, n_genes=1500, umi_len=10, permille_neighbour=150)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    o = parity.oracle_run(Oracle, dict(merge_kind=5, max_cb_merge_ed=max_ed, min_genes_before=3, min_genes_after=10), cb, umi, gene, aux)
    c = parity.gpu_run(dict(merge_kind=capi.MERGE_ALL, max_cb_merge_edit_distance=max_ed, min_genes_before_merge=3, min_genes_after_merge=10),
                       cb, umi, gene, aux)
    parity.compare(o, c)
    mt = c.merge_targets()
    assert int((mt != np.arange(len(mt))).sum()) > 20


# ---------------------------------------------------------------------------------------------------
# whitelists of three parts (the reference's data/barcodes/split_seq and 10x_v2_0_split, configs/split_seq.xml)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("whitelist,poisson", [("split_seq", False), ("10x_v2_0_split", False), ("split_seq", True)])
def test_three_part_whitelist_merge(whitelist, poisson):
    s = SynthStream(n_reads=150_000, whitelist=whitelist, cb_len=24, n_cells=30, n_genes=1500, umi_len=10, permille_neighbour=150)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    path = os.path.join(DATA, whitelist)
    if poisson:
        okw, gkw = _poisson_kw(path, capi.BARCODES_CONST, 3, 10)
    else:
        okw = dict(merge_kind=1, barcodes_kind=1, barcodes_file=path, min_genes_before=3, min_genes_after=10, min_merge_fraction=0.0)
        gkw = dict(merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=capi.BARCODES_CONST, barcodes_file=path, min_genes_before_merge=3,
                   min_genes_after_merge=10, min_merge_fraction=0.0)
    o = parity.oracle_run(Oracle, okw, cb, umi, gene, aux)
    c = parity.gpu_run(gkw, cb, umi, gene, aux)
    parity.compare(o, c)
    mt = c.merge_targets()
    assert int((mt != np.arange(len(mt))).sum()) > 20


def test_indrop_v1_2_whitelist_with_barcodes_of_several_lengths():
    """configs/indrop_v1_2.xml: the first barcode part has 8-11 bases (InDropBarcodesParser splits off the fixed-length
    second part, InDropBarcodesParser.cpp:32-39), so cell barcodes of four lengths live in one container."""
    from dropest_amd.synth import load_whitelist
    p1, p2 = load_whitelist(os.path.join(DATA, "indrop_v1_2"))
    assert {len(x) for x in p1} == {8, 9, 10, 11} and {len(x) for x in p2} == {8}
    rng = np.random.default_rng(21)
    real = [p1[int(i)] + p2[int(j)] for i, j in zip(rng.choice(len(p1), 12, replace=False), rng.choice(len(p2), 12, replace=False))]

    def mutate(s):
        i = int(rng.integers(0, len(s)))
        return s[:i] + str(rng.choice([c for c in "ACGT" if c != s[i]])) + s[i + 1:]
    # every real cell has two mis-read copies of its barcode that see the same molecules (so they merge into it)
    variants = {r: [r] * 8 + [mutate(r), mutate(r)] for r in real}
    reads = []
    for _ in range(20_000):
        r = str(rng.choice(real))
        u = "".join(rng.choice(list("ACGT"), 4)); g = "G%d" % int(rng.integers(0, 12))
        reads.append((str(rng.choice(variants[r])), u, g))
    cb, umi, gene, aux, names = _pack_reads(reads)
    cb, umi, gene, aux = parity.canonical_stream(cb, umi, gene, aux)
    path = os.path.join(DATA, "indrop_v1_2")
    o = parity.oracle_run(Oracle, dict(merge_kind=1, barcodes_kind=0, barcodes_file=path, min_genes_before=3, min_genes_after=10),
                          cb, umi, gene, aux)
    c = parity.gpu_run(dict(merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=capi.BARCODES_INDROP, barcodes_file=path,
                            min_genes_before_merge=3, min_genes_after_merge=10), cb, umi, gene, aux)
    parity.compare(o, c)
    mt = c.merge_targets()
    assert int((mt != np.arange(len(mt))).sum()) >= 12


@pytest.mark.parametrize("with_n", [False, True])
def test_count_matrices_under_other_mark_queries(with_n):
    """ResultsPrinter::save_intron_exon_matrices (-V): the filtered matrix for the queries "e", "i", "BA" (and a few
    more) next to the container's own -L, with UMI counts and with read counts, also over groups rewritten by the
    N-UMI merge."""
    s = SynthStream(n_reads=120_000, n_cells=25, n_genes=400, umi_len=8, permille_intron=250, permille_exon_na=150)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    side = ()
    if with_n:
        umi, side = inject_n(umi, gene, 0.01, 13, 8)
    o = parity.oracle_run(Oracle, dict(min_genes_before=3, min_genes_after=10), cb, umi, gene, aux, side)
    c = parity.gpu_run(dict(min_genes_before_merge=3, min_genes_after_merge=10), cb, umi, gene, aux, side)
    parity.compare(o, c, side)
    total = 0
    for levels in ("e", "i", "BA", "eEBA", "iI", "A"):
        for reads_output in (False, True):
            g, col, v = c.count_matrix_levels(levels, reads_output)
            og, ocol, ov = o.count_matrix_levels(levels, reads_output)
            assert np.array_equal(g.astype(np.uint64), og) and np.array_equal(col.astype(np.uint64), ocol) and np.array_equal(v.astype(np.uint64), ov), levels
            total += len(g)
    g0, c0, v0 = c.count_matrix(filtered=True)
    g1, c1, v1 = c.count_matrix_levels("eEBA")
    assert np.array_equal(g0, g1) and np.array_equal(v0, v1)            # the default query again
    assert total > 10_000


def test_public_mutators_exclude_merge_cells_merge_umis():
    """CellsDataContainer::exclude_cell / merge_cells / merge_umis called directly (CellsDataContainer.cpp:90-109, :209-213),
    between set_initialized and merge_and_filter, against the same calls on the oracle; then the reference's testUMIMerge
    (Tests/TestEstimation.cpp:468-488)."""
    s = SynthStream(n_reads=60_000, n_cells=20, n_genes=120, umi_len=6, permille_neighbour=0)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    okw = dict(min_genes_before=3, min_genes_after=5)
    gkw = dict(min_genes_before_merge=3, min_genes_after_merge=5)
    o = Oracle(**okw); o.add_packed(cb, umi, gene, aux, ()); o.set_initialized()
    c = capi.Context(**gkw); c.push_reads(cb, umi, gene, aux); c.set_initialized()
    real = [int(i) for i in np.nonzero(c.cell_rows()["is_real"])[0]]
    assert len(real) >= 8
    a, b, d, e, f = real[0], real[3], real[5], real[6], real[7]
    # merges (also a chain: d -> b after b received a), an exclusion
    for src, tgt in ((a, b), (d, b)):
        o.merge_cells(src, tgt); c.merge_cells(src, tgt)
    o.exclude_cell(e); c.exclude_cell(e)
    # explicit UMI merge inside one (cell, gene) group: a -> existing target, a -> new UMI, identity pair skipped
    g, u, r, m = c.cell_molecules(f)
    g0 = int(g[0]); mine = [int(x) for x, gg in zip(u, g) if int(gg) == g0]
    assert len(mine) >= 3
    new_code = capi.pack_seq("TTTTTT") if capi.pack_seq("TTTTTT") not in mine else capi.pack_seq("GGGGGG")
    pairs = [(mine[0], mine[1]), (mine[2], new_code), (mine[1], mine[1])]
    c.merge_umis(f, g0, pairs)
    o.merge_umis_explicit(f, "G%d" % g0, {capi.unpack_code(x): capi.unpack_code(y) for x, y in pairs})
    with pytest.raises(capi.DropestError):
        c.merge_umis(f, g0, [(mine[0], mine[1])])            # "Source UMI doesn't belong to the gene"
    o.merge_and_filter(); c.merge_and_filter()
    parity.compare(o, c)
    rows = c.cell_rows()
    assert rows["is_merged"][a] == 1 and rows["is_merged"][d] == 1 and rows["is_excluded"][e] == 1
    assert list(c.merge_targets()) == list(range(c.total_cells_number()))   # the strategy's targets: identity (no -m)

    # testUMIMerge
    reads = [("AAATTAGGTCCA", "AAACCT", "Gene1"), ("AAATTAGGTCCA", "CCCCCT", "Gene1"), ("AAATTAGGTCCA", "AAATTN", "Gene1"),
             ("AAATTAGGTCCA", "ACCCCT", "Gene1")]
    side, index = [], {}

    def code(sq):
        p = capi.pack_seq(sq)
        if p is not None:
            return p
        index.setdefault(sq, len(side)); side.append(sq) if len(side) == index[sq] else None
        return capi.ESCAPE | index[sq]
    cb2 = np.array([code(x[0]) for x in reads], np.uint64); umi2 = np.array([code(x[1]) for x in reads], np.uint64)
    t = capi.Context(min_genes_before_merge=0, min_genes_after_merge=0)
    t.set_side_strings(side)
    t.push_reads(cb2, umi2, np.zeros(4, np.uint32), np.full(4, 2 << 16, np.uint32))
    t.set_initialized()
    t.merge_umis(0, 0, [(code("AAACCT"), code("CCCCCT")), (code("AAATTN"), code("GGGGGG")), (code("ACCCCT"), code("ACCCCT"))])
    g, u, r, m = t.cell_molecules(0)
    got = {capi.unpack_code(x, side): int(y) for x, y in zip(u, r)}
    assert got == {"CCCCCT": 2, "GGGGGG": 1, "ACCCCT": 1}                                       # :484-487


def test_prefetched_raw_matrix_is_the_same_matrix():
    """dropest_prefetch_raw_matrix: cm_raw produced on the second stream equals the one produced on demand; a change of
    the container or another value kind discards the prefetch."""
    s = SynthStream(n_reads=300_000, n_cells=80, n_genes=4000)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    def fresh():
        c = capi.Context(min_genes_before_merge=10, min_genes_after_merge=20)
        c.push_reads(cb, umi, gene, aux); c.set_initialized(); c.merge_and_filter()
        return c
    ref = fresh()
    want = [a.copy() for a in ref.count_matrix_csc(filtered=False)]
    want_reads = [a.copy() for a in ref.count_matrix_csc(filtered=False, reads_output=True)]
    c = fresh()
    c.prefetch_raw_matrix()
    cm = [a.copy() for a in c.count_matrix_csc(filtered=True)]            # work on the main stream in between
    got = [a.copy() for a in c.count_matrix_csc(filtered=False)]
    assert all(np.array_equal(a, b) for a, b in zip(want, got))
    assert all(np.array_equal(a, b) for a, b in zip(cm, ref.count_matrix_csc(filtered=True)))
    # another value kind than the prefetched one: produced on demand
    c.prefetch_raw_matrix()
    got = [a.copy() for a in c.count_matrix_csc(filtered=False, reads_output=True)]
    assert all(np.array_equal(a, b) for a, b in zip(want_reads, got))
    # the container changes after the prefetch: the stale matrix must not be returned
    c.prefetch_raw_matrix()
    victim = int(c.filtered_cells()[0])
    c.exclude_cell(victim); ref.exclude_cell(victim)
    got = [a.copy() for a in c.count_matrix_csc(filtered=False)]
    want2 = ref.count_matrix_csc(filtered=False)
    assert all(np.array_equal(a, b) for a, b in zip(want2, got)) and len(got[0]) == len(want[0]) - 1
    # idempotent over passes
    c.reset_results(); c.set_initialized(); c.merge_and_filter(); c.prefetch_raw_matrix()
    got = [a.copy() for a in c.count_matrix_csc(filtered=False)]
    assert all(np.array_equal(a, b) for a, b in zip(want, got))
