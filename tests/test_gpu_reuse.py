"""Nothing on the path may read device memory it has not written.

The debug allocator of csrc/util.h makes every temporary adversarial: allocations filled with pseudo-random patterns, released
blocks recycled uncleared (stale data of earlier stages -- what a buffer pool does), and dropest_debug_poison_scratch
overwrites every buffer a context keeps across passes (and the pinned staging buffers).  Every observable of a pass must be
independent of all of that.  The large case is the C3 shape (whitelist merge) at 1.5e8 reads on ONE reused context."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from dropest_amd import capi
from dropest_amd.synth import SynthStream

pytestmark = pytest.mark.gpu

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dropest_amd", "data", "barcodes")
DEBUG_VARS = ("DROPEST_POISON_SEED", "DROPEST_DEBUG_POOL", "DROPEST_POISON_ZERO", "DROPEST_POISON_ALLOC")
OLD_PATHS = {"DROPEST_EXACT_INGEST_STATS": "1", "DROPEST_CB_NO_HOT": "1", "DROPEST_SORT": "lsd", "DROPEST_SS_BALLOT_RANK": "1"}


def digests(c):
    rows = c.cell_rows()
    obs = {"cm%d" % j: x for j, x in enumerate(c.count_matrix_csc(filtered=True))}
    obs.update({"raw%d" % j: x for j, x in enumerate(c.count_matrix_csc(filtered=False))})
    obs.update({"row:" + k: rows[k] for k in rows.dtype.names})
    obs["filtered"] = np.array(c.filtered_cells()); obs["targets"] = np.array(c.merge_targets())
    obs["counters"] = np.array(c.global_counters())
    m = c.molecules()
    obs.update({"mol%d" % j: x for j, x in enumerate(m)})
    return {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() for k, v in obs.items()}


def poison(seed):
    n = C.c_uint64()
    assert capi.lib().dropest_debug_poison_scratch(seed, C.byref(n)) == 0
    return n.value


@pytest.fixture(autouse=True)
def clean_env():
    saved = {k: os.environ.get(k) for k in DEBUG_VARS + tuple(OLD_PATHS) + ("DROPEST_DEBUG_REGISTRY",)}
    os.environ["DROPEST_DEBUG_REGISTRY"] = "1"     # every allocation of these tests is tracked: dropest_debug_poison_scratch overwrites what is live
    capi.lib().dropest_debug_refresh()            # (the library reads the debug switches once; this has them read again)
    yield
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    capi.lib().dropest_debug_trim_pool()
    capi.lib().dropest_debug_refresh()


def merge_kw(before=10, after=60):
    return dict(min_genes_before_merge=before, min_genes_after_merge=after, merge_kind=capi.MERGE_REAL_BARCODES,
                barcodes_kind=capi.BARCODES_CONST, barcodes_file=os.path.join(DATA, "10x_aug_2016_split"))


def test_c3_shape_1_5e8_reads_twice_on_one_context_with_different_fills():
    """The C3 shape at 1.5e8 reads, whitelist merge: pass 1 on a fresh context, then every kept buffer and every pinned staging
    buffer overwritten with pattern A, pass 2, overwritten with pattern B, pass 3 -- SHA-256 of every observable equal."""
    dev = SynthStream(n_reads=150_000_000, n_cells=7500, n_genes=30000, umi_len=12, stream_id=3, permille_neighbour=50).generate_device(0)
    c = capi.Context(**merge_kw(20, 100))
    c.push_reads_device(*dev.ptrs, dev.n, adopt=True)
    c.set_initialized(); c.merge_and_filter()
    ref = digests(c)
    assert int(c.cell_rows()["is_merged"].sum()) > 10000
    for seed in (0xA5A5, 0x5A5A):
        c.reset_results()
        assert poison(seed) > 40
        c.set_initialized(); c.merge_and_filter()
        got = digests(c)
        assert got == ref, [k for k in ref if ref[k] != got[k]]
    c.close(); dev.free()


@pytest.mark.parametrize("paths", ["fast", "conservative"])
@pytest.mark.parametrize("mode", ["random-1", "random-2", "recycled", "recycled+random"])
def test_allocator_state_does_not_change_the_result(mode, paths):
    """A 2.4e7-read whitelist merge through fresh contexts: allocations filled with random patterns, blocks recycled uncleared
    from a previous pass of another shape, both at once -- on the fast paths and on the conservative ones (LSD sort, exact
    ingest statistics, no hot list, ballot ranking)."""
    if paths == "conservative":
        os.environ.update(OLD_PATHS)
    dev = SynthStream(n_reads=24_000_000, n_cells=6000, n_genes=20000, umi_len=12, stream_id=11, permille_neighbour=120).generate_device(0)
    other = SynthStream(n_reads=9_000_000, n_cells=900, n_genes=4000, umi_len=8, stream_id=12, permille_neighbour=60).generate_device(0)

    def run(d, env):
        for k in DEBUG_VARS:
            os.environ.pop(k, None)
        os.environ.update(env)
        capi.lib().dropest_debug_refresh()
        c = capi.Context(**merge_kw())
        c.push_reads_device(*d.ptrs, d.n, adopt=True)
        c.set_initialized(); c.merge_and_filter()
        out = digests(c)
        c.close()
        for k in DEBUG_VARS:
            os.environ.pop(k, None)
        capi.lib().dropest_debug_refresh()
        return out
    ref = run(dev, {})
    env = {"random-1": {"DROPEST_POISON_SEED": "1"}, "random-2": {"DROPEST_POISON_SEED": "2"}, "recycled": {"DROPEST_DEBUG_POOL": "1"},
           "recycled+random": {"DROPEST_DEBUG_POOL": "1", "DROPEST_POISON_SEED": "3"}}[mode]
    if "recycled" in mode:   # fill the pool with the blocks of another stream's pass, then of this stream's
        run(other, {"DROPEST_DEBUG_POOL": "1"})
        run(dev, {"DROPEST_DEBUG_POOL": "1"})
    got = run(dev, env)
    assert got == ref, [k for k in ref if ref[k] != got[k]]
    dev.free(); other.free()


def test_no_merge_pass_after_poison_and_other_stream():
    """C2 shape (no CB merge): stream A, kept buffers overwritten, stream B on the same context == B on a fresh context."""
    a = SynthStream(n_reads=12_000_000, n_cells=800, n_genes=15000, stream_id=21).generate_device(0)
    b = SynthStream(n_reads=7_000_000, n_cells=2500, n_genes=30000, stream_id=22).generate_device(0)
    kw = dict(min_genes_before_merge=20, min_genes_after_merge=100)
    c = capi.Context(**kw)
    c.push_reads_device(*a.ptrs, a.n, adopt=True)
    c.set_initialized(); c.merge_and_filter()
    c.clear_reads()
    poison(77)
    c.push_reads_device(*b.ptrs, b.n, adopt=True)
    c.set_initialized(); c.merge_and_filter()
    got = digests(c)
    f = capi.Context(**kw)
    f.push_reads_device(*b.ptrs, b.n, adopt=True)
    f.set_initialized(); f.merge_and_filter()
    ref = digests(f)
    assert got == ref, [k for k in ref if ref[k] != got[k]]
    c.close(); f.close(); a.free(); b.free()
