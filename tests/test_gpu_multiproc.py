"""The sharded runner with one PROCESS per shard (what bench.py launches under torch.distributed.run), on a box with one GPU: the
processes share the device, so the device data of the collectives crosses through POSIX shared memory (DROPEST_SHARD_DATAPLANE=shm,
csrc/shard_run.h) instead of RCCL, which refuses two ranks on one device.  Everything else is the multi-process code of a real run: the
host collectives through the shm mailbox with real peers, the node-shared result buffers registered in every process, the step's sequence
of collectives -- partition, all-to-all(v), key-field agreement, the sharded whitelist merge, N-UMI resolution, device-planned cm_raw,
every shard writing its columns.  Results must equal ONE context over the whole stream."""
import ctypes as C
import multiprocessing as mp
import os
import sys
import tempfile

import numpy as np
import pytest

from dropest_amd import capi
from dropest_amd.multi import cfg_kwargs
from dropest_amd.synth import SynthStream, inject_n

import parity

pytestmark = pytest.mark.gpu
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dropest_amd", "data", "barcodes")


def _worker(rank, world, uid, cfg_kw, side, npz, out, steps, opts=()):
    try:
        os.environ["DROPEST_SHARD_DATAPLANE"] = "shm"
        from dropest_amd import capi as cp
        from dropest_amd.multi import Shard
        L = cp.lib()
        cfg, keep = cp.make_cfg(device=0, **cfg_kw)
        h = C.c_void_p()
        ub = np.frombuffer(uid, np.uint8).copy()
        rc = L.dropest_shard_create(C.byref(cfg), rank, world, ub.ctypes.data, C.byref(h))
        assert rc == 0, L.dropest_last_error()
        sh = Shard(h.value)
        d = np.load(npz)
        n = len(d["cb"])
        lo, hi = n * rank // world, n * (rank + 1) // world
        if side:
            sh.set_side_strings(side)
        for k, v in opts:
            sh.set_option(k, v)
        sh.set_reads(cp.DeviceArrays.from_host(0, d["cb"][lo:hi], d["umi"][lo:hi], d["gene"][lo:hi], d["aux"][lo:hi]), lo)
        for _ in range(steps):
            sh.step()
        if rank == 0:
            cm, raw = sh.matrix(True), sh.matrix(False)
            ms, mt = sh.merged_barcodes()
            np.savez(out, cm_p=cm[0], cm_i=cm[1], cm_x=cm[2], cm_b=cm[3], raw_p=raw[0], raw_i=raw[1], raw_x=raw[2], raw_b=raw[3], ms=ms, mt=mt,
                     moved=np.array([sh.phase_stats()["all_to_all"]["bytes"]]))
        sh.close()
    except BaseException as e:   # noqa: BLE001 (the parent reads the text)
        import traceback
        open(out + ".err%d" % rank, "w").write(traceback.format_exc())
        raise


def run_processes(world, arrays, cfg_kw, side=(), steps=1, opts=()):
    tmp = tempfile.mkdtemp(prefix="dropest_mp_")
    npz, out = os.path.join(tmp, "reads.npz"), os.path.join(tmp, "out.npz")
    np.savez(npz, cb=arrays[0], umi=arrays[1], gene=arrays[2], aux=arrays[3])
    uid = np.random.default_rng(os.getpid()).integers(0, 256, 128, dtype=np.uint8).tobytes()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, uid, cfg_kw, tuple(side), npz, out, steps, tuple(opts))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    errs = [open(out + ".err%d" % r).read() for r in range(world) if os.path.exists(out + ".err%d" % r)]
    assert not errs and all(p.exitcode == 0 for p in procs), "\n".join(errs) or [p.exitcode for p in procs]
    return np.load(out)


def check(got, arrays, cfg_kw, side=()):
    c = capi.Context(**cfg_kw)
    if side:
        c.set_side_strings(side)
    c.push_reads(*arrays)
    c.set_initialized(); c.merge_and_filter()
    rows = c.cell_rows()
    for filt, name in ((True, "cm"), (False, "raw")):
        p, i, x = c.count_matrix_csc(filtered=filt)
        assert np.array_equal(got[name + "_p"].astype(np.uint64), p.astype(np.uint64)), name
        assert np.array_equal(got[name + "_i"], i) and np.array_equal(got[name + "_x"], x), name
    assert [int(b) for b in got["cm_b"]] == [int(rows["barcode"][int(k)]) for k in c.filtered_cells()]
    assert [int(b) for b in got["raw_b"]] == [int(b) for b in rows["barcode"][rows["is_real"].astype(bool)]]
    mt = c.merge_targets()
    src = np.flatnonzero(mt != np.arange(len(mt)))
    want = {int(rows["barcode"][k]): int(rows["barcode"][int(mt[k])]) for k in src}
    assert dict(zip((int(b) for b in got["ms"]), (int(b) for b in got["mt"]))) == want
    assert float(got["moved"][0]) > 0          # reads really crossed between the processes
    c.close()
    return want


@pytest.mark.parametrize("world", [2, 3])
def test_processes_sharing_one_gpu_match_single_context(world):
    """No CB merge, UMIs with N (one global rand() sequence, global first occurrences), two passes on the same shards."""
    s = SynthStream(n_reads=400_000, n_cells=60, n_genes=3000)
    cb, umi, gene, aux = parity.canonical_stream(*s.generate_host())
    umi, side = inject_n(umi, gene, 2e-3, 5, 10)
    kw = cfg_kwargs({"min_before": 10, "min_after": 30})
    got = run_processes(world, (cb, umi, gene, aux), kw, side, steps=2)
    check(got, (cb, umi, gene, aux), kw, side)


def test_processes_whitelist_merge_across_shards():
    """-m with the 10x whitelist over two processes: merge targets on the other process, molecule rows moving between them."""
    s = SynthStream(n_reads=300_000, n_cells=40, n_genes=2000, umi_len=12, permille_neighbour=150)
    arrays = parity.canonical_stream(*s.generate_host())
    kw = cfg_kwargs({"min_before": 3, "min_after": 20, "merge": {"barcodes_kind": capi.BARCODES_CONST, "barcodes_file": os.path.join(DATA, "10x_aug_2016_split")}})
    got = run_processes(2, arrays, kw)
    want = check(got, arrays, kw)
    assert len(want) > 20


@pytest.mark.parametrize("kind", ["simple", "poisson_simple", "all"])
def test_processes_whitelist_free_merges_across_shards(kind):
    """-m / -M without a whitelist and merge_type = all over three processes: the UMI-gene index, the partial pair counts and the replay
    queries cross process borders through the transport's all-to-all and the mailbox (csrc/shard_merge_free.h)."""
    if kind == "simple":   # few UMIs and genes: near-ties, replayed with global cell indices
        s = SynthStream(n_reads=60_000, n_cells=12, n_genes=40, umi_len=3, permille_neighbour=250)
        kw = dict(merge_kind=capi.MERGE_SIMPLE, max_cb_merge_edit_distance=17, min_merge_fraction=0.0, min_genes_before_merge=1, min_genes_after_merge=1)
    else:
        s = SynthStream(n_reads=150_000, whitelist="10x_aug_2016_split", n_cells=30, n_genes=1500, umi_len=8, permille_neighbour=150)
        kw = dict(merge_kind=capi.MERGE_POISSON_SIMPLE if kind == "poisson_simple" else capi.MERGE_ALL, max_cb_merge_edit_distance=2,
                  max_real_merge_prob=1e-3, min_genes_before_merge=3, min_genes_after_merge=10)
    arrays = parity.canonical_stream(*s.generate_host())
    got = run_processes(3, arrays, kw)
    want = check(got, arrays, kw)
    assert len(want) > 20


def test_processes_byte_lists_overflow_falls_back():
    """A list segment of 16 entries per shard: both matrices overflow on both processes; all of them place the columns again in the
    16-bit form (the shared buffer is reopened larger by every process) -- same matrices as one context."""
    s = SynthStream(n_reads=300_000, n_cells=50, n_genes=9000, umi_len=8)
    arrays = parity.canonical_stream(*s.generate_host())
    kw = cfg_kwargs({"min_before": 1, "min_after": 5})
    got = run_processes(2, arrays, kw, steps=2, opts=(("slots_matrix", 0), ("byte_list_cap", 16)))
    check(got, arrays, kw)


@pytest.mark.parametrize("world,chunks", [(2, 4), (3, 3)])
def test_processes_chunked_exchange(world, chunks):
    """The all-to-all in chunks under the partition and the barcode table (shard option exchange_chunks; the default from 2^22 reads per
    shard on): piece k of every process's blocks crosses the shm data plane while piece k + 1 is partitioned, the table takes piece k of
    every source.  Whitelist merge behind it; two passes.  Same matrices and merged barcodes as one context."""
    s = SynthStream(n_reads=300_000, n_cells=40, n_genes=2000, umi_len=12, permille_neighbour=150)
    arrays = parity.canonical_stream(*s.generate_host())
    kw = cfg_kwargs({"min_before": 3, "min_after": 20, "merge": {"barcodes_kind": capi.BARCODES_CONST, "barcodes_file": os.path.join(DATA, "10x_aug_2016_split")}})
    got = run_processes(world, arrays, kw, steps=2, opts=(("exchange_chunks", chunks),))
    want = check(got, arrays, kw)
    assert len(want) > 20

