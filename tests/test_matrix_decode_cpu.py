"""The host decoder of the count matrices' byte form (dropest_amd/csrc/matrix_decode.h: what dropest_count_matrix_csc runs under its
device-to-host copies, and dropest_matrix_bytes_widen with everything already on the host) against a numpy encoder / decoder of the
format include/dropest_amd.h states.  No GPU: the walk is plain host code."""
import ctypes as C

import numpy as np
import pytest

from dropest_amd import capi


def encode(colptr, rows, vals, rng):
    """dropest_matrix_bytes of a CSC matrix; the lists in random order (the device appends to them as it goes)."""
    nnz = len(rows)
    prev = np.empty(nnz, np.int64)
    prev[1:] = rows[:-1]
    starts = colptr[:-1][np.diff(colptr.astype(np.int64)) > 0]
    prev[starts] = -1
    delta = rows.astype(np.int64) - prev
    assert (delta > 0).all()
    d8 = np.where(delta >= 255, 255, delta).astype(np.uint8)
    v8 = np.where(vals >= 255, 255, vals).astype(np.uint8)
    rp = np.flatnonzero(delta >= 255).astype(np.uint32)
    vp = np.flatnonzero(vals >= 255).astype(np.uint32)
    rng.shuffle(rp); rng.shuffle(vp)
    return d8, v8, rp, rows[rp].astype(np.uint32), vp, vals[vp].astype(np.uint32)


def random_matrix(rng, ncols, n_genes, dense_cols, big_value_rate=0.001):
    """Columns of very different lengths: `dense_cols` long ones (thousands of genes: small gaps) among short sparse ones (all gaps listed)."""
    lens = rng.integers(0, 60, ncols)
    lens[rng.choice(ncols, dense_cols, replace=False)] = rng.integers(300, min(n_genes, 6000), dense_cols)
    colptr = np.zeros(ncols + 1, np.uint32)
    colptr[1:] = np.cumsum(lens)
    rows = np.concatenate([np.sort(rng.choice(n_genes, int(k), replace=False)) for k in lens] + [np.zeros(0, np.int64)]).astype(np.uint32)
    vals = rng.geometric(0.3, len(rows)).astype(np.uint32)
    big = rng.random(len(rows)) < big_value_rate
    vals[big] = rng.integers(255, 1 << 20, int(big.sum()))
    return colptr, rows, vals


def widen(colptr, d8, v8, rp, rr, vp, vv, misalign=0):
    m = capi.MatrixBytes()
    m.ncols, m.nnz = len(colptr) - 1, len(d8)
    keep = [np.ascontiguousarray(a) for a in (colptr, d8, v8, rp, rr, vp, vv)]
    m.colptr, m.row_delta, m.value = (a.ctypes.data for a in keep[:3])
    m.n_row_listed, m.row_listed_pos, m.row_listed_row = len(rp), keep[3].ctypes.data, keep[4].ctypes.data
    m.n_value_listed, m.value_listed_pos, m.value_listed_value = len(vp), keep[5].ctypes.data, keep[6].ctypes.data
    ro = np.full(m.nnz + 16, 0xDEADBEEF, np.uint32)[misalign:misalign + m.nnz]     # (different alignments of the two outputs: the
    vo = np.full(m.nnz + 16, 0xDEADBEEF, np.uint32)[3:3 + m.nnz]                    # non-temporal path needs them equal modulo 32 bytes)
    st = capi.lib().dropest_matrix_bytes_widen(C.byref(m), ro.ctypes.data, vo.ctypes.data)
    return st, ro, vo


@pytest.mark.parametrize("seed,ncols,dense,misalign", [(1, 3000, 40, 3), (2, 200, 200, 0), (3, 50000, 0, 3), (4, 1, 1, 5), (5, 7000, 300, 11)])
def test_widen_equals_the_matrix_it_encodes(seed, ncols, dense, misalign):
    rng = np.random.default_rng(seed)
    colptr, rows, vals = random_matrix(rng, ncols, 30000 if seed != 5 else 70000, dense)
    st, ro, vo = widen(colptr, *encode(colptr, rows, vals, rng), misalign=misalign)
    assert st == 0
    assert np.array_equal(ro, rows) and np.array_equal(vo, vals)


def test_widen_of_an_empty_matrix_and_of_empty_columns():
    z8, z32 = np.zeros(0, np.uint8), np.zeros(0, np.uint32)
    st, ro, vo = widen(np.zeros(5, np.uint32), z8, z8, z32, z32, z32, z32)
    assert st == 0 and len(ro) == 0


def test_a_listed_entry_that_does_not_stand_on_a_255_is_refused():
    rng = np.random.default_rng(7)
    colptr, rows, vals = random_matrix(rng, 500, 30000, 20)
    d8, v8, rp, rr, vp, vv = encode(colptr, rows, vals, rng)
    bad = np.flatnonzero(d8 != 255)[:1].astype(np.uint32)
    st, _, _ = widen(colptr, d8, v8, np.concatenate([rp, bad]), np.concatenate([rr, rows[bad]]), vp, vv)
    assert st != 0 and b"255" in capi.lib().dropest_last_error()
    st, _, _ = widen(colptr, d8, v8, rp, rr, np.concatenate([vp, [np.uint32(len(d8))]]).astype(np.uint32), np.concatenate([vv, [np.uint32(1)]]).astype(np.uint32))
    assert st != 0


def test_scalar_walk_equals_the_vector_walk(monkeypatch):
    """DROPEST_DECODE / DROPEST_DECODE_NT are read once per process: children run the same matrix through every walk the CPU has."""
    import subprocess, sys, os
    code = ("import numpy as np, sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_matrix_decode_cpu as t\n"
            "rng = np.random.default_rng(11); colptr, rows, vals = t.random_matrix(rng, 4000, 30000, 60)\n"
            "st, ro, vo = t.widen(colptr, *t.encode(colptr, rows, vals, rng))\n"
            "assert st == 0 and np.array_equal(ro, rows) and np.array_equal(vo, vals)\n") % (os.path.dirname(os.path.dirname(__file__)), os.path.dirname(__file__))
    for env in ({"DROPEST_DECODE": "scalar"}, {"DROPEST_DECODE": "avx2", "DROPEST_DECODE_NT": "1"}, {"DROPEST_DECODE": "avx2", "DROPEST_DECODE_NT": "0"},
                {"DROPEST_DECODE_NT": "0"}, {"DROPEST_DECODE_NT": "1"}, {"DROPEST_DECODE_THREADS": "1"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]


def test_workers_that_nap_after_claiming_a_slice_never_make_a_slice_count_twice(monkeypatch):
    """a worker that claimed a slice, lost its CPU, and then stored 'somebody walks it' over the rescuer's 'done'
    made the slice count twice -- the job finished one slice early.  Small slices + a nap between the claim and the walk: every slice
    is rescued by the caller while its worker sleeps; the library checks that every slice was counted exactly once."""
    monkeypatch.setenv("DROPEST_DECODE_SLICE", "256")
    monkeypatch.setenv("DROPEST_DECODE_TEST_DELAY_US", "300")
    rng = np.random.default_rng(21)
    for _ in range(6):
        colptr, rows, vals = random_matrix(rng, 3000, 30000, 30)
        st, ro, vo = widen(colptr, *encode(colptr, rows, vals, rng))
        assert st == 0, capi.lib().dropest_last_error()
        assert np.array_equal(ro, rows) and np.array_equal(vo, vals)


# ---- a matrix that rides on another one's rows (round 6: cm on cm_raw; dropest_matrix_rider_widen) ----
def make_rider(rng, colptr, rows, vals, p_col=0.7, p_keep=0.95, big_rate=0.002):
    """A subset of the base's columns in another order, a column's entries a subset of the base's with values <= the base's (a few beyond 254)."""
    ncols = len(colptr) - 1
    chosen = np.flatnonzero(rng.random(ncols) < p_col)
    rng.shuffle(chosen)
    value = np.zeros(len(rows), np.uint8)
    out_begin = np.full(ncols, 0xFFFFFFFF, np.uint32); out_count = np.zeros(ncols, np.uint32)
    want_rows, want_vals, lpos, lval = [], [], [], []
    at = 0
    for c in chosen:
        b, e = int(colptr[c]), int(colptr[c + 1])
        keep = rng.random(e - b) < p_keep
        v = np.minimum(vals[b:e], np.maximum(1, rng.integers(1, 400, e - b))).astype(np.uint32)
        big = rng.random(e - b) < big_rate
        v[big] = rng.integers(255, 1 << 20, int(big.sum()))
        v[~keep] = 0
        value[b:e] = np.where(v >= 255, 255, v)
        kept = np.flatnonzero(v)
        out_begin[c] = at; out_count[c] = len(kept)
        for j in np.flatnonzero(v[kept] >= 255):
            lpos.append(at + j); lval.append(v[kept][j])
        want_rows.append(rows[b:e][kept]); want_vals.append(v[kept])
        at += len(kept)
    perm = rng.permutation(len(lpos))
    return (value, out_begin, out_count, at, np.array(lpos, np.uint32)[perm], np.array(lval, np.uint32)[perm],
            np.concatenate(want_rows + [np.zeros(0, np.uint32)]).astype(np.uint32), np.concatenate(want_vals + [np.zeros(0, np.uint32)]).astype(np.uint32))


def rider_widen(colptr, d8, rows, value, out_begin, out_count, nnz_out, lpos, lval):
    L = capi.lib()
    L.dropest_matrix_rider_widen.restype = C.c_int
    L.dropest_matrix_rider_widen.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    m = capi.MatrixBytes()
    keep = [np.ascontiguousarray(a) for a in (colptr, d8, rows, value, out_begin, out_count, lpos, lval)]
    m.ncols, m.nnz = len(colptr) - 1, len(d8)
    m.colptr, m.row_delta = keep[0].ctypes.data, keep[1].ctypes.data
    ro = np.full(nnz_out + 16, 0xDEADBEEF, np.uint32); vo = np.full(nnz_out + 16, 0xDEADBEEF, np.uint32)
    st = L.dropest_matrix_rider_widen(C.byref(m), keep[2].ctypes.data, keep[3].ctypes.data, keep[4].ctypes.data, keep[5].ctypes.data, nnz_out, len(lpos),
                                      keep[6].ctypes.data, keep[7].ctypes.data, ro.ctypes.data, vo.ctypes.data)
    return st, ro, vo


@pytest.mark.parametrize("seed,ncols,dense", [(31, 3000, 40), (32, 300, 300), (33, 40000, 0), (34, 2, 2)])
def test_rider_takes_its_rows_from_the_base(seed, ncols, dense):
    rng = np.random.default_rng(seed)
    colptr, rows, vals = random_matrix(rng, ncols, 30000, dense)
    d8 = encode(colptr, rows, vals, rng)[0]
    value, ob, oc, nnz_out, lpos, lval, want_r, want_v = make_rider(rng, colptr, rows, vals)
    st, ro, vo = rider_widen(colptr, d8, rows, value, ob, oc, nnz_out, lpos, lval)
    assert st == 0, capi.lib().dropest_last_error()
    assert np.array_equal(ro[:nnz_out], want_r) and np.array_equal(vo[:nnz_out], want_v)
    assert (ro[nnz_out:] == 0xDEADBEEF).all() and (vo[nnz_out:] == 0xDEADBEEF).all()      # nothing written behind the matrix
    if nnz_out:      # a column announced with one entry too few is refused
        c = int(np.flatnonzero(oc)[0])
        oc2 = oc.copy(); oc2[c] -= 1
        st, _, _ = rider_widen(colptr, d8, rows, value, ob, oc2, nnz_out, lpos, lval)
        assert st != 0 and b"announced" in capi.lib().dropest_last_error()


def test_rider_scalar_walk_equals_the_vector_walk():
    import subprocess, sys, os
    code = ("import numpy as np, sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_matrix_decode_cpu as t\n"
            "rng = np.random.default_rng(41); colptr, rows, vals = t.random_matrix(rng, 4000, 30000, 60)\n"
            "d8 = t.encode(colptr, rows, vals, rng)[0]\n"
            "value, ob, oc, n, lp, lv, wr, wv = t.make_rider(rng, colptr, rows, vals)\n"
            "st, ro, vo = t.rider_widen(colptr, d8, rows, value, ob, oc, n, lp, lv)\n"
            "assert st == 0 and np.array_equal(ro[:n], wr) and np.array_equal(vo[:n], wv)\n") % (os.path.dirname(os.path.dirname(__file__)), os.path.dirname(__file__))
    for env in ({"DROPEST_DECODE": "scalar"}, {}, {"DROPEST_DECODE_THREADS": "1"}, {"DROPEST_DECODE_SLICE": "256", "DROPEST_DECODE_TEST_DELAY_US": "200"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
