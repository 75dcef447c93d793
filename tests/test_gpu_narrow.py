"""The narrow CSC form (dropest_count_matrix_csc_narrow: 16-bit rows and values + an exact overflow list) must widen to exactly
the 32-bit matrices of dropest_count_matrix_csc -- what ResultsPrinter::create_matrix builds (ResultsPrinter.cpp:433-442)."""
import numpy as np
import pytest

from dropest_amd import capi
from dropest_amd.synth import SynthStream

import parity

pytestmark = pytest.mark.gpu


def both_forms(c, reads_output=False, prefetch=False):
    for filt in (True, False):
        if prefetch and not filt:
            c.prefetch_raw_matrix(reads_output, narrow=True)
        n = c.count_matrix_csc_narrow(filtered=filt, reads_output=reads_output)
        got = [x.copy() for x in capi.Context.widen(n)]
        n_ovf = len(n[3])
        want = [x.copy() for x in c.count_matrix_csc(filtered=filt, reads_output=reads_output)]
        for g, w in zip(got, want):
            assert g.dtype == w.dtype and np.array_equal(g, w), (filt, reads_output)
        yield filt, n_ovf, want


@pytest.mark.parametrize("reads_output", [False, True])
@pytest.mark.parametrize("prefetch", [False, True])
def test_narrow_equals_wide_on_a_c2_shape(reads_output, prefetch):
    s = SynthStream(n_reads=3_000_000, n_cells=300, n_genes=20000)
    c = capi.Context(min_genes_before_merge=20, min_genes_after_merge=100)
    c.push_reads(*parity.canonical_stream(*s.generate_host()))
    c.set_initialized(); c.merge_and_filter()
    assert c.narrow_matrix_possible()
    seen = list(both_forms(c, reads_output, prefetch))
    assert all(len(w[1]) > 100000 for _, _, w in seen)
    # a wide call after a narrow prefetch, and the other way round, still give the right form
    c.prefetch_raw_matrix(reads_output, narrow=True)
    wide = [x.copy() for x in c.count_matrix_csc(filtered=False, reads_output=reads_output)]
    c.prefetch_raw_matrix(reads_output, narrow=False)
    nar = capi.Context.widen(c.count_matrix_csc_narrow(filtered=False, reads_output=reads_output))
    assert all(np.array_equal(a, b) for a, b in zip(wide, nar))
    c.close()


def test_values_beyond_16_bits_go_through_the_overflow_list():
    """One cell whose gene 0 has 70 000 molecules (one read each) and gene 1 has one molecule of 66 000 reads: both matrices, UMI
    counts and read counts, carry entries > 65534; a third gene stays small."""
    P = capi.pack_seq
    rng = np.random.default_rng(5)
    n0 = 70_000
    umis0 = np.arange(n0, dtype=np.uint64) | np.uint64(1 << 20)           # 70 000 distinct 10-base codes (sentinel bit 20)
    cb = np.full(n0 + 66_000 + 3, P("ACGTACGTACGTACGT"), np.uint64)
    umi = np.concatenate([umis0, np.full(66_000, P("TTTTTTTTTT"), np.uint64), np.array([P("AAAAAAAAAC"), P("AAAAAAAAAG"), P("AAAAAAAAAT")], np.uint64)])
    gene = np.concatenate([np.zeros(n0, np.uint32), np.ones(66_000, np.uint32), np.full(3, 2, np.uint32)])
    aux = np.full(len(cb), 0 | (2 << 16), np.uint32)
    perm = rng.permutation(len(cb))
    cb, umi, gene, aux = parity.canonical_stream(cb[perm], umi[perm], gene[perm], aux[perm])
    c = capi.Context(min_genes_before_merge=1, min_genes_after_merge=1)
    c.push_reads(cb, umi, gene, aux)
    c.set_initialized(); c.merge_and_filter()
    for reads_output, n_over in ((False, 1), (True, 2)):
        for filt, n_ovf, want in both_forms(c, reads_output):
            assert n_ovf == n_over, (reads_output, filt, n_ovf)
            assert int(want[2].max()) == (70_000 if not reads_output else 70_000)
    c.close()


def test_gene_ids_beyond_16_bits_refuse_the_narrow_form():
    cb = np.array([capi.pack_seq("ACGTACGTACGTACGT")] * 2, np.uint64)
    umi = np.array([capi.pack_seq("ACGTACGTAC"), capi.pack_seq("ACGTACGTAA")], np.uint64)
    gene = np.array([3, 70_000], np.uint32); aux = np.array([2 << 16, 2 << 16], np.uint32)
    c = capi.Context(min_genes_before_merge=0, min_genes_after_merge=0)
    c.push_reads(cb, umi, gene, aux)
    c.set_initialized(); c.merge_and_filter()
    assert not c.narrow_matrix_possible()
    with pytest.raises(capi.DropestError):
        c.count_matrix_csc_narrow(filtered=True)
    assert len(c.count_matrix_csc(filtered=True)[1]) == 2
    c.close()


def test_empty_container_narrow():
    e = capi.Context(min_genes_before_merge=0, min_genes_after_merge=0)
    e.set_initialized(); e.merge_and_filter()
    n = e.count_matrix_csc_narrow(filtered=True)
    assert len(n[1]) == 0 and len(n[3]) == 0
    e.close()


# ---- the byte form (dropest_count_matrix_csc_bytes): one byte of row delta, one byte of value, two exact lists -----------------------------
def _bytes_equal_wide(c, reads_output=False, prefetch=False):
    out = []
    for filt in (True, False):
        if prefetch and not filt:
            c.prefetch_raw_matrix(reads_output, form=2)
        m = c.count_matrix_csc_bytes(filtered=filt, reads_output=reads_output)
        got = [x.copy() for x in c.widen_bytes(m)]
        listed = (int(m.n_row_listed), int(m.n_value_listed), int(m.nnz))
        want = [x.copy() for x in c.count_matrix_csc(filtered=filt, reads_output=reads_output)]
        for g, w in zip(got, want):
            assert g.dtype == w.dtype and np.array_equal(g, w), (filt, reads_output)
        out.append((filt, listed, want))
    return out


@pytest.mark.parametrize("reads_output", [False, True])
@pytest.mark.parametrize("prefetch", [False, True])
def test_byte_form_equals_wide_on_a_c2_shape(reads_output, prefetch):
    s = SynthStream(n_reads=3_000_000, n_cells=300, n_genes=20000)
    c = capi.Context(min_genes_before_merge=20, min_genes_after_merge=100)
    c.push_reads(*parity.canonical_stream(*s.generate_host()))
    c.set_initialized(); c.merge_and_filter()
    seen = _bytes_equal_wide(c, reads_output, prefetch)
    for _, (n_rows, n_vals, nnz), w in seen:
        assert nnz > 100000 and n_rows < nnz // 20 and n_vals < nnz // 20    # the lists stay short: that is the point of the form
    # the three forms after one another, prefetches of another form in between
    c.prefetch_raw_matrix(reads_output, form=2)
    wide = [x.copy() for x in c.count_matrix_csc(filtered=False, reads_output=reads_output)]
    c.prefetch_raw_matrix(reads_output, narrow=True)
    b = [x.copy() for x in c.widen_bytes(c.count_matrix_csc_bytes(filtered=False, reads_output=reads_output))]
    nar = capi.Context.widen(c.count_matrix_csc_narrow(filtered=False, reads_output=reads_output))
    assert all(np.array_equal(x, y) for x, y in zip(wide, b)) and all(np.array_equal(x, y) for x, y in zip(wide, nar))
    c.close()


def test_byte_form_lists_large_gaps_large_counts_and_columns_longer_than_a_round():
    """Gene ids up to 200 000 (no 16-bit form possible), gaps of every size around 255, counts around 255, a cell with more than 256
    genes (several rounds of the emit kernel: the delta crosses a round boundary), cells with a single gene."""
    P = capi.pack_seq
    rows = []
    cells = ["ACGTACGTACGTACG" + x for x in "ACGT"]
    genes0 = [0, 253, 507, 762, 1018, 1019, 1500, 70_000, 70_254, 70_509, 200_000]          # gaps 1(-1->0), 253, 254, 255, 256, 1, ...
    counts0 = [1, 253, 254, 255, 256, 300, 2, 254, 255, 1, 70_000]
    for g, n in zip(genes0, counts0):
        rows += [(cells[0], int(u), g) for u in range(n)]
    rows += [(cells[1], 0, 123_456)]                                                       # one gene, far away: listed row, first entry
    rows += [(cells[2], 0, 254)]                                                           # first delta exactly 255 -> listed
    rows += [(cells[2], 1, 253 + 255)]                                                     # ... and a delta of 254 after a listed row
    rng = np.random.default_rng(11)
    many = np.sort(rng.choice(150_000, size=900, replace=False))
    rows += [(cells[3], int(k % 7), int(g)) for k, g in enumerate(many)]
    cb = np.array([P(r[0]) for r in rows], np.uint64)
    umi = np.array([np.uint64(r[1]) | np.uint64(1 << 34) for r in rows], np.uint64)        # 17-base codes: room for 70 000 distinct UMIs
    gene = np.array([r[2] for r in rows], np.uint32)
    aux = np.full(len(rows), 2 << 16, np.uint32)
    perm = rng.permutation(len(rows))
    # (gene ids stay as given: first-seen renumbering would close the gaps this test is about)
    c = capi.Context(min_genes_before_merge=0, min_genes_after_merge=0)
    c.push_reads(cb[perm], umi[perm], gene[perm], aux[perm])
    c.set_initialized(); c.merge_and_filter()
    assert not c.narrow_matrix_possible()
    for reads_output in (False, True):
        seen = _bytes_equal_wide(c, reads_output)
        for filt, (n_rows, n_vals, nnz), want in seen:
            assert nnz == len(genes0) + 1 + 2 + 900
            assert n_rows >= 6 and n_vals == 5, (n_rows, n_vals)      # 255 (twice), 256, 300, 70 000 are listed; 253 / 254 are not
    c.close()


def test_byte_form_of_an_empty_container():
    c = capi.Context()
    c.set_initialized(); c.merge_and_filter()
    m = c.count_matrix_csc_bytes(filtered=True)
    assert m.nnz == 0 and m.ncols == 0
    c.close()


@pytest.mark.parametrize("with_n", [False, True])
@pytest.mark.parametrize("form", [0, 1, 2])
def test_announced_cm_raw_prefetch_changes_nothing(form, with_n):
    """dropest_set_raw_matrix_prefetch: the container starts cm_raw's prefetch by itself -- at the end of set_initialized when
    merge_and_filter cannot change the matrix, after it otherwise (UMIs with N are merged there) -- and every form of both matrices is
    what a context without the announcement gives; the announcement survives a second pass over the same reads."""
    from dropest_amd.synth import inject_n
    s = SynthStream(n_reads=1_500_000, n_cells=200, n_genes=8000)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    side = ()
    if with_n:
        umi, side = inject_n(umi, gene, 1e-3, 5, 10)
    def make(announce):
        c = capi.Context(min_genes_before_merge=20, min_genes_after_merge=100)
        if side:
            c.set_side_strings(side)
        c.push_reads(cb, umi, gene, aux)
        if announce:
            c.set_raw_matrix_prefetch(form)
        c.set_initialized(); c.merge_and_filter()
        return c
    ref = make(False)
    want = {f: [x.copy() for x in ref.count_matrix_csc(filtered=f)] for f in (True, False)}
    c = make(True)
    for rounds in range(2):
        for filt in (True, False):
            if form == 2:
                got = c.widen_bytes(c.count_matrix_csc_bytes(filtered=filt))
            elif form == 1:
                got = capi.Context.widen(c.count_matrix_csc_narrow(filtered=filt))
            else:
                got = c.count_matrix_csc(filtered=filt)
            assert all(np.array_equal(a, b) for a, b in zip(got, want[filt])), (form, with_n, filt, rounds)
        c.reset_results(); c.set_initialized(); c.merge_and_filter()
    c.close(); ref.close()


def test_push_reads_gather_equals_one_push():
    """dropest_push_reads_gather: runs that follow one another in the stream, handed over in one call, are the stream."""
    s = SynthStream(n_reads=400_000, n_cells=80, n_genes=3000)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    cuts = [0, 1, 1, 70_001, 200_000, 399_999, 400_000]                     # (an empty run, runs of one read)
    a = capi.Context(min_genes_before_merge=10, min_genes_after_merge=30)
    a.push_reads(cb, umi, gene, aux)
    b = capi.Context(min_genes_before_merge=10, min_genes_after_merge=30)
    b.push_reads_gather([(cb[i:j], umi[i:j], gene[i:j], aux[i:j]) for i, j in zip(cuts[:-1], cuts[1:])][:3])
    b.push_reads_gather([(cb[i:j], umi[i:j], gene[i:j], aux[i:j]) for i, j in zip(cuts[:-1], cuts[1:])][3:])
    for c in (a, b):
        c.set_initialized(); c.merge_and_filter()
    for filt in (True, False):
        assert all(np.array_equal(x, y) for x, y in zip(a.count_matrix_csc(filtered=filt), b.count_matrix_csc(filtered=filt)))
    assert np.array_equal(a.filtered_cells(), b.filtered_cells())
    a.close(); b.close()
