"""dropest_merge_apply (MergeStrategyBase::merge_inited second loop + reassign, MergeStrategyBase.cpp:30-82) is host arithmetic: its
worker-thread path (no target is itself merged away: the steps commute) against the serial restatement, on the CPU -- with and
without chains, self targets, excluded cells and targets that are excluded themselves."""
import ctypes as C
import os

import numpy as np
import pytest

from dropest_amd import capi


def _apply(n_cells, order, target, reads, umis):
    L = capi.lib()
    L.dropest_merge_apply.restype = C.c_int
    L.dropest_merge_apply.argtypes = [C.c_uint64, C.c_uint64] + [C.c_void_p] * 6
    r, u = reads.copy(), umis.copy()
    final = np.zeros(n_cells, np.uint32)
    excl = np.zeros(n_cells, np.uint8)
    rc = L.dropest_merge_apply(n_cells, len(order), order.ctypes.data, target.ctypes.data, r.ctypes.data, u.ctypes.data, final.ctypes.data, excl.ctypes.data)
    assert rc == 0, L.dropest_last_error().decode()
    return r, u, final, excl


def _case(rng, n_cells, chains):
    order = rng.permutation(n_cells).astype(np.uint32)[: int(n_cells * 0.9)]
    sinks = rng.choice(n_cells, max(2, n_cells // 50), replace=False)              # "whitelist" cells: targets
    target = np.full(len(order), -1, np.int64)
    kind = rng.integers(0, 10, len(order))
    is_sink = np.zeros(n_cells, bool); is_sink[sinks] = True
    for i, c in enumerate(order):
        if is_sink[c]:
            target[i] = c if kind[i] < 8 else -1                                    # a real barcode maps to itself, or is excluded
        elif kind[i] < 7:
            target[i] = int(rng.choice(sinks))
    if chains:                                                                      # some targets merge on themselves: order matters
        idx = np.flatnonzero(is_sink[order])
        for i in rng.choice(idx, max(1, len(idx) // 4), replace=False):
            target[i] = int(rng.choice(sinks))
    reads = rng.integers(1, 1000, n_cells).astype(np.int32)
    umis = rng.integers(1, 500, n_cells).astype(np.int32)
    return order, target, reads, umis


@pytest.mark.parametrize("chains", [False, True])
@pytest.mark.parametrize("n_cells", [50, 3000, 200_000])
def test_worker_thread_path_equals_the_serial_loop(monkeypatch, n_cells, chains):
    rng = np.random.default_rng(n_cells + chains)
    order, target, reads, umis = _case(rng, n_cells, chains)
    monkeypatch.setenv("DROPEST_SERIAL_MERGE_ORDER", "1")
    want = _apply(n_cells, order, target, reads, umis)
    monkeypatch.delenv("DROPEST_SERIAL_MERGE_ORDER")
    monkeypatch.setenv("DROPEST_PARALLEL_MERGE_ORDER_MIN", "8")
    got = _apply(n_cells, order, target, reads, umis)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    assert (want[2] != np.arange(n_cells)).sum() > n_cells // 3                     # cells were merged
    assert want[3].sum() > 0                                                        # and excluded


def test_bad_indices_still_raise(monkeypatch):
    monkeypatch.setenv("DROPEST_PARALLEL_MERGE_ORDER_MIN", "1")
    order = np.array([0, 1, 9], np.uint32)
    target = np.array([1, 1, -1], np.int64)
    L = capi.lib()
    L.dropest_merge_apply.restype = C.c_int
    L.dropest_merge_apply.argtypes = [C.c_uint64, C.c_uint64] + [C.c_void_p] * 6
    z = np.zeros(4, np.int32); f = np.zeros(4, np.uint32); e = np.zeros(4, np.uint8)
    assert L.dropest_merge_apply(4, 3, order.ctypes.data, target.ctypes.data, z.ctypes.data, z.copy().ctypes.data, f.ctypes.data, e.ctypes.data) != 0
