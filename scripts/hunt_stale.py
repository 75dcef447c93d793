"""Hunt for reads of device memory that was never written.

One stream, one configuration, run several times through FRESH contexts under different states of the debug allocator
(csrc/util.h): plain; every allocation filled with a pseudo-random pattern (two seeds); released blocks recycled uncleared
(DROPEST_DEBUG_POOL: stale data of earlier stages, the adversary that found the round-2 dependence); one context reused with
its kept buffers overwritten between passes.  Every observable is digested; a digest that differs from the plain run is
bisected over the allocation ordinals of the pass (DROPEST_POISON_ZERO=a:b zero-fills the allocations numbered [a, b)): the
smallest set of allocations whose clearing restores the plain result names the buffer that is read before it is written.

    python scripts/hunt_stale.py                 # default shapes (C3-like, 1.0-1.6e8 reads, whitelist merge)
    READS=150000000 CELLS=7500 SHAPES=3 python scripts/hunt_stale.py
"""
import ctypes as C
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
os.environ["DROPEST_ALLOC_TRACE"] = "1"
os.environ["DROPEST_DEBUG_REGISTRY"] = "1"
import numpy as np

from dropest_amd import capi
from dropest_amd.synth import SynthStream

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dropest_amd", "data", "barcodes")
DEBUG_VARS = ("DROPEST_POISON_SEED", "DROPEST_DEBUG_POOL", "DROPEST_POISON_ZERO", "DROPEST_POISON_ALLOC")
# the conservative paths of the library (what scripts/soak_paths.py compares the fast paths with): a dependence may sit in either
OLD_PATHS = {"DROPEST_EXACT_INGEST_STATS": "1", "DROPEST_CB_NO_HOT": "1", "DROPEST_SORT": "lsd", "DROPEST_SS_BALLOT_RANK": "1"}


def next_ordinal():
    n = C.c_uint64()
    assert capi.lib().dropest_debug_alloc_ordinal(C.byref(n)) == 0
    return n.value


def site(ordinal):
    buf = C.create_string_buffer(512)
    capi.lib().dropest_debug_alloc_site(ordinal, buf, 512)
    return buf.value.decode()


def observables(c):
    rows = c.cell_rows()
    out = {"cm%d" % j: x for j, x in enumerate(c.count_matrix_csc(filtered=True))}
    out.update({"raw%d" % j: x for j, x in enumerate(c.count_matrix_csc(filtered=False))})
    out.update({"row:" + k: rows[k] for k in rows.dtype.names})
    out["filtered"] = np.array(c.filtered_cells())
    out["targets"] = np.array(c.merge_targets())
    out["counters"] = np.array(c.global_counters())
    return out


def digest(obs):
    return {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()[:16] for k, v in obs.items()}


def one_pass(dev, kw, env, ctx=None, keep=False):
    """A pass under `env` (debug variables).  Returns (digests, first ordinal, next ordinal, context if keep)."""
    for k in DEBUG_VARS:
        os.environ.pop(k, None)
    os.environ.update(env)
    capi.lib().dropest_debug_refresh()
    o0 = next_ordinal()
    c = ctx or capi.Context(**kw)
    if ctx is None:
        c.push_reads_device(*dev.ptrs, dev.n, adopt=True)
    c.set_initialized(); c.merge_and_filter()
    d = digest(observables(c))
    o1 = next_ordinal()
    if not keep:
        c.close()
    for k in DEBUG_VARS:
        os.environ.pop(k, None)
    return d, o0, o1, (c if keep else None)


def differing(a, b):
    return [k for k in a if a[k] != b[k]]


def bisect(dev, kw, env, plain, o_len, warm):
    """Smallest prefix / suffix windows of allocation ordinals whose zero-filling makes the pass under `env` equal `plain`."""
    def ok(a, b):   # relative ordinals [a, b) cleared
        if warm:
            warm()
        base = next_ordinal()
        e = dict(env); e["DROPEST_POISON_ZERO"] = "%d:%d" % (base + a, base + b)
        d, o0, o1, _ = one_pass(dev, kw, e)
        assert o0 == base, (o0, base)
        return not differing(plain, d), o1 - o0
    good, n = ok(0, o_len + 64)
    print("   clearing every allocation restores the result:", good, "(%d allocations)" % n, flush=True)
    if not good:
        return None
    lo, hi = 0, o_len   # smallest hi such that clearing [0, hi) is enough
    while lo < hi:
        mid = (lo + hi) // 2
        if ok(0, mid)[0]: hi = mid
        else: lo = mid + 1
    end = lo
    lo, hi = 0, end     # largest start such that clearing [start, end) is enough
    while lo < hi:
        mid = (lo + hi + 1) // 2
        if ok(mid, end)[0]: lo = mid
        else: hi = mid - 1
    return lo, end


def main():
    rng = np.random.default_rng(int(os.environ.get("SEED", "7")))
    n_shapes = int(os.environ.get("SHAPES", "3"))
    found = 0
    for it in range(n_shapes):
        n = int(os.environ.get("READS", rng.integers(int(os.environ.get("NMIN", "100000000")), int(os.environ.get("NMAX", "160000000")))))
        shape = dict(n_reads=n, n_cells=int(os.environ.get("CELLS", rng.integers(50, 20000))), n_genes=int(rng.integers(200, 40000)),
                     umi_len=int(rng.choice([8, 10, 12])), stream_id=int(rng.integers(1, 1000)), permille_neighbour=int(rng.integers(0, 200)))
        kw = dict(min_genes_before_merge=int(rng.integers(1, 30)), min_genes_after_merge=int(rng.integers(30, 120)),
                  merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=capi.BARCODES_CONST, barcodes_file=os.path.join(DATA, "10x_aug_2016_split"))
        old = bool(it % 2) if os.environ.get("PATHS", "both") == "both" else os.environ["PATHS"] == "old"
        for k in OLD_PATHS:
            os.environ.pop(k, None)
        if old:
            os.environ.update(OLD_PATHS)
        print("shape", it, shape, {k: v for k, v in kw.items() if k.startswith("min")}, "conservative paths" if old else "fast paths", flush=True)
        dev = SynthStream(**shape).generate_device(0)
        t0 = time.time()
        plain, o0, o1, _ = one_pass(dev, kw, {})
        o_len = o1 - o0
        print("  plain: %d allocations, %.1f s" % (o_len, time.time() - t0), flush=True)
        again, *_ = one_pass(dev, kw, {})
        if differing(plain, again):
            print("  NOT DETERMINISTIC without any poison:", differing(plain, again), flush=True)
            found += 1

        def warm_pool():   # a pass that leaves its blocks in the pool
            one_pass(dev, kw, {"DROPEST_DEBUG_POOL": "1"})

        modes = [("random fill, seed 1", {"DROPEST_POISON_SEED": "1"}, None), ("random fill, seed 2", {"DROPEST_POISON_SEED": "2"}, None),
                 ("recycled blocks (cold pool)", {"DROPEST_DEBUG_POOL": "1"}, None), ("recycled blocks (warm pool)", {"DROPEST_DEBUG_POOL": "1"}, warm_pool),
                 ("recycled blocks + random fill of fresh ones", {"DROPEST_DEBUG_POOL": "1", "DROPEST_POISON_SEED": "3"}, warm_pool)]
        for name, env, warm in modes:
            if warm:
                warm()
            d, a0, a1, _ = one_pass(dev, kw, env)
            bad = differing(plain, d)
            print("  %-45s %s" % (name, "equal" if not bad else "DIFFERS in " + ", ".join(bad)), flush=True)
            if bad:
                found += 1
                win = bisect(dev, kw, env, plain, a1 - a0, warm)
                if win:
                    print("   culprit allocations (relative ordinals %d..%d):" % (win[0], win[1] - 1), flush=True)
                    # sites of a fresh traced pass under the same mode
                    if warm:
                        warm()
                    base = next_ordinal()
                    one_pass(dev, kw, env)
                    for o in range(win[0], win[1]):
                        print("     #%d  %s" % (o, site(base + o)), flush=True)
            capi.lib().dropest_debug_trim_pool()
        # one context, two passes, kept buffers overwritten in between
        for seed in (11, 12):
            d1, _, _, c = one_pass(dev, kw, {}, keep=True)
            c.reset_results()
            nb = C.c_uint64()
            assert capi.lib().dropest_debug_poison_scratch(seed, C.byref(nb)) == 0
            d2, _, _, _ = one_pass(dev, kw, {}, ctx=c, keep=True)
            c.close()
            bad = differing(plain, d2) or differing(plain, d1)
            print("  %-45s %s" % ("reused context, %d kept blocks overwritten (seed %d)" % (nb.value, seed), "equal" if not bad else "DIFFERS in " + ", ".join(bad)), flush=True)
            found += bool(bad)
        dev.free()
    print("dependences found:", found, flush=True)
    return 1 if found else 0


if __name__ == "__main__":
    raise SystemExit(main())
