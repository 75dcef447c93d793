"""dropest_count_matrix_csc hands out the dgCMatrix slots i / x as 32-bit arrays (ResultsPrinter.cpp:433-442).  A large matrix crosses PCIe
in the byte form and is widened into the slots by host threads while the copy runs (csrc/matrix_decode.h); a matrix whose sparse columns
overflow the byte form's row list is emitted as 32-bit arrays after all.  Either way the arrays must be exactly what the direct copy gives."""
import numpy as np
import pytest

from dropest_amd import capi
from dropest_amd.synth import SynthStream

import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _rider_on_small_matrices(monkeypatch):
    """cm rides on cm_raw's rows from 2^26 entries on (round 6): these tests take that path on their small matrices as well."""
    monkeypatch.setenv("DROPEST_RIDER_MIN_NNZ", "0")


def matrices(c, wire, reads_output=False):
    c.set_matrix_wire(wire)
    return {f: [x.copy() for x in c.count_matrix_csc(filtered=f, reads_output=reads_output)] for f in (True, False)}


def same(a, b):
    return all(np.array_equal(x, y) and x.dtype == y.dtype for f in (True, False) for x, y in zip(a[f], b[f]))


@pytest.mark.parametrize("reads_output", [False, True])
@pytest.mark.parametrize("announce", [False, True])
def test_wire_equals_direct_on_a_c2_shape(reads_output, announce):
    s = SynthStream(n_reads=4_000_000, n_cells=400, n_genes=20000)
    c = capi.Context(min_genes_before_merge=20, min_genes_after_merge=100)
    c.push_reads(*parity.canonical_stream(*s.generate_host()))
    if announce:
        c.set_raw_matrix_prefetch(0, reads_output)
    c.set_initialized(); c.merge_and_filter()
    wire = matrices(c, True, reads_output)
    assert len(wire[False][1]) >= (1 << 18)              # large enough to take the byte way
    direct = matrices(c, False, reads_output)
    assert same(wire, direct)
    # a second pass on the same context (kept buffers, kept events), the announced prefetch in flight while cm is made
    c.set_matrix_wire(True)
    c.reset_results(); c.set_initialized(); c.merge_and_filter()
    assert same(matrices(c, True, reads_output), direct)
    c.close()


def sparse_stream(n_cells, genes_per_cell, n_genes, seed):
    """Many small cells: a few dozen genes out of tens of thousands each, one read per molecule -- every row gap is far beyond 254."""
    rng = np.random.default_rng(seed)
    n = n_cells * genes_per_cell
    cell = np.repeat(np.arange(n_cells, dtype=np.uint64), genes_per_cell)
    cb = (np.uint64(1) << np.uint64(32)) | (cell * np.uint64(2654435761) & np.uint64(0xFFFFFFFF))     # distinct 16-base codes (sentinel bit 32)
    gene = rng.integers(0, n_genes, n).astype(np.uint32)
    umi = (np.uint64(1) << np.uint64(20)) | rng.integers(0, 1 << 20, n).astype(np.uint64)
    aux = np.full(n, 2 << 16, np.uint32)
    perm = rng.permutation(n)
    return cb[perm], umi[perm], gene[perm], aux[perm]


def test_sparse_cells_overflow_the_row_list_and_fall_back(monkeypatch):
    """150 000 cells of ~40 genes out of 60 000: 6e6 entries, nearly all of them with a row gap beyond 254 -- more than the row list of the
    byte form takes (max(2^20, nnz / 8)).  The public byte form refuses; dropest_count_matrix_csc (what the facade's ResultsPrinter calls)
    still delivers, with and without an announced prefetch, equal to the direct copy."""
    cb, umi, gene, aux = sparse_stream(150_000, 40, 60_000, 3)
    for announce in (False, True):
        c = capi.Context(min_genes_before_merge=10, min_genes_after_merge=20)
        c.push_reads(cb, umi, gene, aux)
        if announce:
            c.set_raw_matrix_prefetch(0)
        c.set_initialized(); c.merge_and_filter()
        wire = matrices(c, True)
        assert len(wire[False][1]) > 5_000_000
        c.set_matrix_wire(True)
        with pytest.raises(capi.DropestError):
            c.count_matrix_csc_bytes(filtered=False)
        assert same(wire, matrices(c, False))
        c.close()


def test_a_forced_small_row_list_falls_back_on_a_dense_matrix(monkeypatch):
    s = SynthStream(n_reads=3_000_000, n_cells=300, n_genes=20000)
    c = capi.Context(min_genes_before_merge=20, min_genes_after_merge=100)
    c.push_reads(*parity.canonical_stream(*s.generate_host()))
    c.set_initialized(); c.merge_and_filter()
    direct = matrices(c, False)
    monkeypatch.setenv("DROPEST_MATRIX_ROW_LIST_CAP", "7")
    assert same(matrices(c, True), direct)
    c.set_raw_matrix_prefetch(0)
    c.set_matrix_wire(True)
    c.reset_results(); c.set_initialized(); c.merge_and_filter()
    assert same(matrices(c, True), direct)
    c.close()


@pytest.mark.parametrize("kind", ["plain", "reads", "max_cells", "merge"])
def test_cm_rides_on_cm_raw(kind, monkeypatch):
    """Round 6: with cm_raw on its way in the byte form, cm crosses PCIe as one byte per entry of cm_raw (its value in cm, 0 = not in cm) and the
    host threads take the rows from cm_raw's deltas (k_misc.h: emit_values_on_rows_kernel, matrix_decode.h: widen_derived).  The slots must be the
    direct copy's: UMI counts, read counts (values beyond 254 are listed), a -C cut (most real cells are no column of cm), after a whitelist merge."""
    import os
    monkeypatch.setenv("DROPEST_RIDER_MIN_NNZ", "0")      # (the library takes the rider from 2^26 entries on: here on a small matrix)
    reads_output = kind == "reads"
    kw = dict(min_genes_before_merge=20, min_genes_after_merge=100)
    if kind == "max_cells":
        kw["max_cells"] = 150
    if kind == "merge":
        data = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dropest_amd", "data", "barcodes")
        kw.update(merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=capi.BARCODES_CONST, barcodes_file=os.path.join(data, "10x_aug_2016_split"))
        s = SynthStream(n_reads=4_000_000, n_cells=400, n_genes=20000, umi_len=12, permille_neighbour=100)
    else:
        s = SynthStream(n_reads=4_000_000, n_cells=400, n_genes=20000, reads_per_molecule=40 if reads_output else 4)
    c = capi.Context(**kw)
    c.push_reads(*parity.canonical_stream(*s.generate_host()))
    c.set_profiling(True)
    for _ in range(2):      # (twice: kept buffers, a rider's job settled before cm_raw's slot is used again)
        c.set_raw_matrix_prefetch(0, reads_output)
        c.set_matrix_wire(True)
        c.reset_results(); c.set_initialized(); c.merge_and_filter()
        c.prefetch_raw_matrix(reads_output=reads_output, form=0)
        cm = [x.copy() for x in c.count_matrix_csc(filtered=True, reads_output=reads_output)]
        raw = [x.copy() for x in c.count_matrix_csc(filtered=False, reads_output=reads_output)]
        assert c.kernel_stats().get("count:cm_rides_on_cm_raw", {"launches": 0})["launches"] >= 1
        direct = matrices(c, False, reads_output)
        assert all(np.array_equal(x, y) for x, y in zip(cm, direct[True])) and all(np.array_equal(x, y) for x, y in zip(raw, direct[False]))
        assert len(cm[1]) > 50_000 and (not reads_output or int(cm[2].max()) > 254)
        if kind == "max_cells":
            assert len(cm[0]) - 1 == 150 and len(raw[0]) - 1 > 300
    c.close()
