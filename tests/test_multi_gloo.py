"""world_size-2 / -3 tests of the N > 1 path on CPU over gloo.

The orchestration of a sharded pass lives in the library (dropest_amd/csrc/shard_run.h) and needs GPUs; what every shard
decides on the HOST -- who owns a barcode, the global column order of the two matrices from the all-gathered table of real
cells, the sequential application of merge targets over the global compare_cells order -- is exposed through the C-ABI
without a device (dropest_owner_of, dropest_plan_columns, dropest_merge_apply).  Here 2 / 3 gloo ranks each run the CPU
oracle on the reads they OWN, all-gather their tables with torch.distributed, and must arrive -- every rank identically --
at what ONE oracle container over the whole stream produces."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dropest_amd import capi
from dropest_amd.synth import SynthStream
from oracle import Oracle

import parity

CFG = {"min_before": 5, "min_after": 10}
STREAM = dict(n_reads=60_000, n_cells=24, n_genes=600, umi_len=8)


def mix64(x):
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(30); x *= np.uint64(0xbf58476d1ce4e5b9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94d049bb133111eb)
        x ^= x >> np.uint64(31)
    return x


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def all_gather_rows(rows):
    """(k, w) int64 per rank -> concatenation over ranks (padded all_gather: gloo has no variable-size one)."""
    world = dist.get_world_size()
    k = torch.tensor([rows.shape[0]], dtype=torch.int64)
    ks = [torch.zeros_like(k) for _ in range(world)]
    dist.all_gather(ks, k)
    ks = [int(x.item()) for x in ks]
    pad = np.zeros((max(ks + [1]), rows.shape[1]), np.int64)
    pad[:rows.shape[0]] = rows
    mine = torch.from_numpy(pad)
    bufs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine)
    return np.concatenate([b.numpy()[:n] for b, n in zip(bufs, ks)])


def shard_table(rank, world, max_cells):
    """What one shard contributes: the oracle over the reads this rank owns (in stream order), one table row per real cell."""
    cb, umi, gene, aux = parity.canonical_stream(*SynthStream(**STREAM).generate_host())
    mine = np.flatnonzero((mix64(cb.copy()) % np.uint64(world)).astype(np.int64) == rank)
    o = Oracle(merge_kind=0, min_genes_before=CFG["min_before"], min_genes_after=CFG["min_after"])
    o.add_packed(cb[mine], umi[mine], gene[mine], aux[mine])
    o.set_initialized(); o.merge_and_filter()
    orows = o.cell_rows()   # merged, excluded, real, n_genes, req_genes, req_umis, total_reads, total_umis
    _, first = np.unique(cb[mine], return_index=True)
    first = np.sort(first)                                   # local cell id k <-> k-th first occurrence among the owned reads
    real = np.flatnonzero(orows[:, 2] == 1)
    table = np.stack([cb[mine][first[real]].astype(np.int64), mine[first[real]].astype(np.int64),      # barcode, GLOBAL first ordinal
                      orows[real, 3], orows[real, 4], orows[real, 5], orows[real, 7], np.full(len(real), rank)], axis=1).astype(np.int64)
    return table


def _worker(rank, world, port, path, max_cells):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        G = all_gather_rows(shard_table(rank, world, max_cells))
        cols = (G[:, 0].astype(np.uint64), G[:, 1].astype(np.uint64), G[:, 2], G[:, 3], G[:, 4], G[:, 5])
        cm = capi.plan_columns(*cols, filtered=True, min_after=CFG["min_after"], max_cells=max_cells)
        raw = capi.plan_columns(*cols, filtered=False, min_after=CFG["min_after"])
        # every rank must hold the same plan: compare with rank 0's
        mine = np.concatenate([[len(cm)], cm, raw]).astype(np.int64)
        plans = all_gather_rows(mine.reshape(-1, 1))
        assert np.array_equal(plans.reshape(world, -1)[rank], plans.reshape(world, -1)[0])
        if rank == 0:
            np.savez(path, cm_bc=G[cm, 0], cm_len=G[cm, 3], cm_owner=G[cm, 6], raw_bc=G[raw, 0], raw_len=G[raw, 2], raw_owner=G[raw, 6])
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("max_cells", [-1, 7])
@pytest.mark.parametrize("world", [2, 3])
def test_global_column_plan_matches_single_container(world, max_cells, tmp_path):
    path = str(tmp_path / "plan.npz")
    mp.spawn(_worker, args=(world, _free_port(), path, max_cells), nprocs=world, join=True)
    got = np.load(path)
    cb, umi, gene, aux = parity.canonical_stream(*SynthStream(**STREAM).generate_host())
    o = Oracle(merge_kind=0, min_genes_before=CFG["min_before"], min_genes_after=CFG["min_after"], max_cells=max_cells)
    o.add_packed(cb, umi, gene, aux)
    o.set_initialized(); o.merge_and_filter()
    orows = o.cell_rows()
    want_cm = [capi.pack_seq(o.cell_barcode(int(i))) for i in o.filtered_cells()]
    assert [int(x) for x in got["cm_bc"]] == want_cm and len(want_cm) > (5 if max_cells < 0 else 6)
    assert [int(x) for x in got["cm_len"]] == [int(orows[int(i), 4]) for i in o.filtered_cells()]      # column length = requested genes
    real_ids = np.flatnonzero(orows[:, 2] == 1)
    assert [int(x) for x in got["raw_bc"]] == [capi.pack_seq(o.cell_barcode(int(i))) for i in real_ids]
    assert [int(x) for x in got["raw_len"]] == [int(orows[int(i), 3]) for i in real_ids]
    # the columns really come from different shards, and each from the shard that owns its barcode
    assert len(set(int(x) for x in got["raw_owner"])) == world
    assert all(capi.lib().dropest_owner_of(int(b), world) == int(r) for b, r in zip(got["raw_bc"], got["raw_owner"]))


def test_owner_function_matches_library():
    codes = np.array([capi.pack_seq(s) for s in ("ACGTACGTACGTACGT", "TTTTTTTTTTTTTTTT", "A", "GATTACA")], np.uint64)
    for n in (1, 2, 3, 8):
        want = [capi.lib().dropest_owner_of(int(c), n) for c in codes]
        assert list((mix64(codes.copy()) % np.uint64(n)).astype(int)) == want


def _merge_worker(rank, world, port, path):
    """Every rank holds a slice of the targets (those of the cells it owns); after the all-gather each applies the whole
    sequence (MergeStrategyBase::merge_inited second loop) and must end with the same final targets as one process."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        n = 400
        sizes = rng.integers(1, 50, n)
        target = np.where(rng.random(n) < 0.15, -1, rng.integers(0, n, n))        # -1 = exclude; chains and self targets included
        reads, umis = rng.integers(1, 1000, n), rng.integers(1, 500, n)
        owner = np.arange(n) % world
        mine = np.flatnonzero(owner == rank)
        rows = all_gather_rows(np.stack([mine, target[mine]], axis=1).astype(np.int64))
        t = np.zeros(n, np.int64); t[rows[:, 0]] = rows[:, 1]
        order = np.lexsort((np.arange(n), sizes)).astype(np.uint32)               # ascending size: the compare_cells order of this toy
        final, excl, r, u = capi.merge_apply(order, t[order], reads, umis)
        if rank == 0:
            np.savez(path, final=final, excl=excl, r=r, u=u)
        every = all_gather_rows(np.concatenate([final, excl, r, u]).astype(np.int64).reshape(-1, 1)).reshape(world, -1)
        assert all(np.array_equal(every[k], every[0]) for k in range(world))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_merge_application_is_identical_on_every_rank(world, tmp_path):
    path = str(tmp_path / "m.npz")
    mp.spawn(_merge_worker, args=(world, _free_port(), path), nprocs=world, join=True)
    got = np.load(path)
    rng = np.random.default_rng(5)
    n = 400
    sizes = rng.integers(1, 50, n)
    target = np.where(rng.random(n) < 0.15, -1, rng.integers(0, n, n))
    reads, umis = rng.integers(1, 1000, n), rng.integers(1, 500, n)
    order = np.lexsort((np.arange(n), sizes)).astype(np.uint32)
    final, excl, r, u = capi.merge_apply(order, target[order], reads, umis)
    assert np.array_equal(final, got["final"]) and np.array_equal(excl, got["excl"]) and np.array_equal(r, got["r"]) and np.array_equal(u, got["u"])
    assert int((final != np.arange(n)).sum()) > 50 and int(excl.sum()) > 20
    assert np.all(r >= reads) and int(r.sum()) > int(reads.sum())                 # Stats::merge ADDS the source's counters to the target (Stats.cpp:29-43)
