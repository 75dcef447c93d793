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
