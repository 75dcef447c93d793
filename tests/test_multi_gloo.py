"""world_size-2 test of the sharded (multi-GPU) orchestration over gloo on CPU tensors.

dropest_amd/multi.py takes the engine that does the per-shard compute as a parameter; in production that is the HIP
path (GpuEngine).  Here a numpy + CPU-oracle engine stands in (TEST ONLY), so that the partition / all-to-all /
global ordering / matrix gather logic is exercised without a GPU, and the assembled result is compared with one
oracle run over the whole stream."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dropest_amd import capi
from dropest_amd.multi import ShardedRun
from dropest_amd.synth import SynthStream
from oracle import Oracle

CFG = {"min_before": 5, "min_after": 10}
STREAM = dict(n_reads=60_000, n_cells=24, n_genes=600, umi_len=8)


def mix64(x):
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(30); x *= np.uint64(0xbf58476d1ce4e5b9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94d049bb133111eb)
        x ^= x >> np.uint64(31)
    return x


def oracle_tables(cb, umi, gene, aux):
    """Runs the oracle on raw-id reads; returns per-cell rows and both matrices with RAW gene ids."""
    o = Oracle(merge_kind=0, min_genes_before=CFG["min_before"], min_genes_after=CFG["min_after"])
    o.add_packed(cb, umi, gene, aux)
    o.set_initialized(); o.merge_and_filter()
    raw_of = np.array([int(o.gene_name(i)[1:]) for i in range(o.n_genes)], np.int64)
    mats = {}
    for filt in (True, False):
        g, c, v = o.count_matrix(filtered=filt)
        g = raw_of[g.astype(np.int64)] if len(g) else np.zeros(0, np.int64)
        order = np.lexsort((g, c.astype(np.int64)))          # column-major, raw gene id ascending in a column
        g, c, v = g[order], c.astype(np.int64)[order], v.astype(np.int64)[order]
        ncols = len(o.filtered_cells()) if filt else o.n_real
        colptr = np.concatenate([[0], np.cumsum(np.bincount(c, minlength=ncols))]).astype(np.int64)
        mats[filt] = (colptr, g, v)
    return o, mats


class CpuEngine:
    """TEST-ONLY stand-in for GpuEngine: numpy for the data movement, the CPU oracle for the per-shard container."""

    def generate(self, stream, first, n):
        cb, umi, gene, aux = stream.generate_host(first, n)
        return [torch.from_numpy(cb.view(np.int64).copy()), torch.from_numpy(umi.view(np.int64).copy()),
                torch.from_numpy(gene.view(np.int32).copy()), torch.from_numpy(aux.view(np.int32).copy())]

    def partition(self, reads, n_parts):
        cb = reads[0].numpy().view(np.uint64)
        owner = (mix64(cb.copy()) % np.uint64(n_parts)).astype(np.int64)
        order = np.argsort(owner, kind="stable")
        out = [r[torch.from_numpy(order)] for r in reads] + [torch.from_numpy(order.astype(np.int32))]
        return out, [int(x) for x in np.bincount(owner, minlength=n_parts)]

    def ingest(self, reads):
        self.reads = reads
        return np.array([2 ** 63, 0, 0, 0, 0, 0], np.uint64)      # neutral: the oracle container has no key layout

    def set_ingest_summary(self, summary):
        pass

    def initialize(self):
        return self._rows()

    def finalize(self):
        return self._rows()

    def _rows(self):
        reads = self.reads
        cb = reads[0].numpy().view(np.uint64); umi = reads[1].numpy().view(np.uint64)
        gene = reads[2].numpy().view(np.uint32); aux = reads[3].numpy().view(np.uint32)
        self.o, self.mats = oracle_tables(cb, umi, gene, aux)
        orows = self.o.cell_rows()
        _, first = np.unique(cb, return_index=True)
        first = np.sort(first)                                   # cell id k <-> k-th first occurrence
        rows = np.zeros(self.o.n_cells, capi.CELL_ROW_DTYPE)
        rows["barcode"] = cb[first]; rows["first_read"] = first
        rows["n_genes"] = orows[:, 3]; rows["requested_genes"] = orows[:, 4]; rows["requested_umis"] = orows[:, 5]
        rows["total_reads"] = orows[:, 6]; rows["total_umis"] = orows[:, 7]; rows["is_real"] = orows[:, 2]
        keep = orows[:, 3] >= CFG["min_before"]
        return np.nonzero(keep)[0].astype(np.uint64), rows[keep]

    def matrix(self, filtered, as_tensors=True):
        colptr, g, v = self.mats[filtered]
        return colptr.astype(np.uint32), torch.from_numpy(g.astype(np.int32)), torch.from_numpy(v.astype(np.int32))

    def filtered_ids(self):
        return self.o.filtered_cells()

    def assemble(self, src, dst, ln, rows, vals, total):
        r = torch.empty(total, dtype=torch.int32); v = torch.empty(total, dtype=torch.int32)
        for s, d, l in zip(src, dst, ln):
            r[d:d + l] = rows[s:s + l]; v[d:d + l] = vals[s:s + l]
        return r, v

    def register_shared(self, buf):
        pass

    def unregister_shared(self, buf):
        pass

    def write_columns(self, src, dst, ln, rows, vals, buf):
        host, cap = buf["host"], buf["cap"]
        r = rows.numpy().view(np.uint32); v = vals.numpy().view(np.uint32)
        for s, d, l in zip(src, dst, ln):
            host[d:d + l] = r[s:s + l]; host[cap + d:cap + d + l] = v[s:s + l]

    def to_numpy_u32(self, t, slot=0):
        return t.numpy().view(np.uint32)

    def take(self, tensor, positions):
        return tensor.numpy()[np.asarray(positions, np.int64)].astype(np.int64)

    def set_profiling(self, on):
        pass

    def kernel_stats(self):
        return {}


def _worker(rank, world, port, result_path, output="shm"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stream = SynthStream(**STREAM)
        per = STREAM["n_reads"] // world
        run = ShardedRun(stream, rank, world, 0, per, dict(CFG, output=output), dist, engine=CpuEngine())
        cm, cm_raw, cols = run.step()
        if rank == 0:
            np.savez(result_path, cm_p=cm[0], cm_i=cm[1], cm_x=cm[2], cm_cols=cm[3],
                     raw_p=cm_raw[0], raw_i=cm_raw[1], raw_x=cm_raw[2], raw_cols=cm_raw[3])
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("world,output", [(2, "shm"), (3, "shm"), (2, "gather")])
def test_sharded_run_matches_single_container(world, output, tmp_path):
    path = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(world, _free_port(), path, output), nprocs=world, join=True)
    got = np.load(path)
    # reference: ONE oracle container over the whole stream
    stream = SynthStream(**STREAM)
    n = (STREAM["n_reads"] // world) * world
    cb, umi, gene, aux = stream.generate_host(0, n)
    o, mats = oracle_tables(cb, umi, gene, aux)
    for filt, pre in ((True, "cm"), (False, "raw")):
        colptr, g, v = mats[filt]
        assert np.array_equal(got[pre + "_p"].astype(np.int64), colptr)
        assert np.array_equal(got[pre + "_i"].astype(np.int64), g)
        assert np.array_equal(got[pre + "_x"].astype(np.int64), v)
    want_cols = [capi.pack_seq(o.cell_barcode(int(i))) for i in o.filtered_cells()]
    assert [int(x) for x in got["cm_cols"]] == want_cols
    real_ids = np.nonzero(o.cell_rows()[:, 2])[0]
    assert [int(x) for x in got["raw_cols"]] == [capi.pack_seq(o.cell_barcode(int(i))) for i in real_ids]
    assert len(want_cols) > 5


def test_owner_function_matches_library():
    codes = np.array([capi.pack_seq(s) for s in ["ACGTACGTACGTACGT", "TTTTTTTTTTTTTTTT", "AAAACCCCGGGGTTTT", "GATTACA"]], np.uint64)
    for n in (1, 2, 3, 8):
        want = [capi.lib().dropest_owner_of(int(c), n) for c in codes]
        assert list((mix64(codes.copy()) % np.uint64(n)).astype(int)) == want
