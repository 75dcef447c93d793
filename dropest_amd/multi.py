"""Sharded runs over several MI355X: ctypes binding of the C++ runner (csrc/shard_run.h, dropest_shard_* in
include/dropest_amd.h).  All orchestration -- partition by owner, the RCCL all-to-all, key-field agreement, the whitelist
CB merge across shards, N-UMI resolution against the one global rand() sequence, global column order, every shard writing
its columns into node-shared host memory -- is in the library; this module only creates the shards and hands out views.

  ShardedRun   one process per GPU (bench.py under torch.distributed.run): the RCCL unique id of the run is made on rank 0
               and broadcast with torch.distributed; torch carries nothing else.
  ShardGroup   all shards inside this process, one host thread each (what the C++ facade does when it owns N GPUs; tests put
               several shards on ONE device).
"""
import ctypes as C

import numpy as np

from . import capi


def cfg_kwargs(cfg):
    """bench / test style config dict -> keyword arguments of capi.make_cfg."""
    m = cfg.get("merge")
    kw = dict(min_genes_before_merge=cfg["min_before"], min_genes_after_merge=cfg["min_after"],
              gene_match_levels=cfg.get("levels", "eEBA"), max_cells=cfg.get("max_cells", -1))
    if m:
        kw.update(merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=m["barcodes_kind"], barcodes_file=m["barcodes_file"],
                  min_merge_fraction=m.get("min_merge_fraction", 0.2))
    return kw


class Shard:
    """One dropest_shard (handle owned here)."""

    def __init__(self, handle):
        self.L = capi.lib()
        self.h = C.c_void_p(handle)
        self.ctx = capi.Context(_borrowed=self.L.dropest_shard_ctx(self.h))
        self._reads = None

    def _chk(self, rc):
        if rc != 0:
            raise capi.DropestError(rc, self.L.dropest_last_error().decode())

    def close(self):
        if self.h:
            self.L.dropest_shard_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_reads(self, arrays, first_ordinal):
        """arrays: capi.DeviceArrays resident on this shard's GPU (kept alive here); its first read is stream ordinal
        `first_ordinal`."""
        self._reads = arrays
        self._chk(self.L.dropest_shard_set_reads_device(self.h, *arrays.ptrs, arrays.n, first_ordinal))

    def push_reads(self, cb, umi, gene, aux, first_ordinal):
        """One batch from host memory (see dropest_shard_push_reads: equal batch lengths, equally spaced first ordinals)."""
        arrs = [np.ascontiguousarray(a, dt) for a, dt in ((cb, np.uint64), (umi, np.uint64), (gene, np.uint32), (aux, np.uint32))]
        self._chk(self.L.dropest_shard_push_reads(self.h, *[a.ctypes.data for a in arrs], len(arrs[0]), int(first_ordinal)))

    def set_side_strings(self, strings):
        self.ctx.set_side_strings(strings)

    def set_umi_qualities(self, qual, lengths=None):
        """qual: uint8 [n_reads of this shard, quality_length], in the order of its reads; lengths: per read, when the strings differ in length."""
        qual = np.ascontiguousarray(qual, np.uint8)
        assert qual.ndim == 2
        if lengths is None:
            self._chk(self.L.dropest_shard_set_umi_qualities(self.h, qual.ctypes.data, qual.shape[1], qual.shape[0]))
        else:
            lengths = np.ascontiguousarray(lengths, np.uint8)
            assert lengths.shape == (qual.shape[0],)
            self._chk(self.L.dropest_shard_set_umi_qualities_var(self.h, qual.ctypes.data, qual.shape[1], lengths.ctypes.data, qual.shape[0]))

    def step(self):
        self._chk(self.L.dropest_shard_step(self.h))

    def set_option(self, key, value):
        self._chk(self.L.dropest_shard_set_option(self.h, key.encode(), int(value)))

    def matrix(self, filtered):
        """(colptr u64[ncols + 1], rowidx u32[nnz], values u32[nnz], column barcodes u64[ncols]) of the GLOBAL matrix: views of
        library-owned memory, valid until the next step (meaningful on shard 0)."""
        ncols, nnz = C.c_uint64(), C.c_uint64()
        p, r, v, b = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._chk(self.L.dropest_shard_matrix(self.h, int(filtered), C.byref(ncols), C.byref(nnz), C.byref(p), C.byref(r), C.byref(v), C.byref(b)))

        def view(ptr, n, ct, dt):
            if not n or not ptr.value:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,))
        return (view(p, ncols.value + 1, C.c_uint64, np.uint64), view(r, nnz.value, C.c_uint32, np.uint32),
                view(v, nnz.value, C.c_uint32, np.uint32), view(b, ncols.value, C.c_uint64, np.uint64))

    def matrix_form(self, filtered):
        """0 = 32-bit arrays, 1 = 16-bit, 2 = byte form only, 3 = byte form + the 32-bit slots widened inside the step."""
        f = C.c_int32()
        self._chk(self.L.dropest_shard_matrix_form(self.h, int(filtered), C.byref(f)))
        return f.value

    def matrix_narrow(self, filtered):
        """(colptr u64, rowidx u16, values u16, column barcodes u64, overflow_pos u64, overflow_val u32) of the GLOBAL matrix in the
        narrow form the step wrote (dropest_shard_matrix_narrow); raises DropestError when the step produced the 32-bit form."""
        ncols, nnz, novf = C.c_uint64(), C.c_uint64(), C.c_uint64()
        p, r, v, b, op, ov = (C.c_void_p() for _ in range(6))
        self._chk(self.L.dropest_shard_matrix_narrow(self.h, int(filtered), C.byref(ncols), C.byref(nnz), C.byref(p), C.byref(r), C.byref(v), C.byref(b),
                                                      C.byref(novf), C.byref(op), C.byref(ov)))

        def view(ptr, n, ct, dt):
            if not n or not ptr.value:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,))
        return (view(p, ncols.value + 1, C.c_uint64, np.uint64), view(r, nnz.value, C.c_uint16, np.uint16), view(v, nnz.value, C.c_uint16, np.uint16),
                view(b, ncols.value, C.c_uint64, np.uint64), view(op, novf.value, C.c_uint64, np.uint64), view(ov, novf.value, C.c_uint32, np.uint32))

    def matrix_bytes(self, filtered):
        """ShardBytes of the GLOBAL matrix in the byte form the step wrote (dropest_shard_matrix_bytes); raises DropestError when
        the shard option byte_matrix is off."""
        m, b = capi.MatrixBytes(), C.c_void_p()
        self._chk(self.L.dropest_shard_matrix_bytes(self.h, int(filtered), C.byref(m), C.byref(b)))
        bc = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_uint64)), shape=(m.ncols,)) if m.ncols and b.value else np.zeros(0, np.uint64)
        return ShardBytes(self.L, m, bc)

    def merged_barcodes(self):
        n = C.c_uint64()
        self._chk(self.L.dropest_shard_merged_barcodes(self.h, C.byref(n), None, None))
        s = np.zeros(n.value, np.uint64); t = np.zeros(n.value, np.uint64)
        if n.value:
            self._chk(self.L.dropest_shard_merged_barcodes(self.h, C.byref(n), s.ctypes.data, t.ctypes.data))
        return s, t

    def phase_stats(self):
        n = C.c_uint32()
        self._chk(self.L.dropest_shard_phase_stats(self.h, C.byref(n), None))
        arr = (capi.KernelStat * max(1, n.value))()
        self._chk(self.L.dropest_shard_phase_stats(self.h, C.byref(n), arr))
        return {arr[i].name.decode(): {"steps": arr[i].launches, "ms": arr[i].ms, "bytes": arr[i].bytes} for i in range(n.value)}


class ShardBytes:
    """A global matrix in the byte form (dropest_matrix_bytes + the column barcodes); widen() decodes it on host threads."""

    def __init__(self, lib, m, col_barcodes):
        self.L, self.m, self.col_barcodes = lib, m, col_barcodes
        self.ncols, self.nnz = int(m.ncols), int(m.nnz)
        self.n_row_listed, self.n_value_listed = int(m.n_row_listed), int(m.n_value_listed)

    def __len__(self):
        return 4

    def __getitem__(self, i):
        if i == 3:
            return self.col_barcodes
        return self.widen()[i]

    def widen(self):
        if not hasattr(self, "_wide"):
            colptr = (np.ctypeslib.as_array(C.cast(self.m.colptr, C.POINTER(C.c_uint32)), shape=(self.ncols + 1,)).astype(np.uint64)
                      if self.ncols else np.zeros(1, np.uint64))
            rows, vals = np.zeros(self.nnz, np.uint32), np.zeros(self.nnz, np.uint32)
            if self.nnz and self.L.dropest_matrix_bytes_widen(C.byref(self.m), rows.ctypes.data, vals.ctypes.data) != 0:
                raise capi.DropestError(-1, self.L.dropest_last_error().decode())
            self._wide = (colptr, rows, vals, self.col_barcodes)
        return self._wide


def widen_shard_matrix(m):
    """(colptr, rowidx u32, values u32, column barcodes) from what ShardedRun.step returns (byte form, narrow 6-tuple or wide 4-tuple)."""
    if isinstance(m, ShardBytes):
        return m.widen()
    if len(m) == 4:
        return m
    colptr, r16, v16, bc, opos, oval = m
    vals = v16.astype(np.uint32)
    vals[opos.astype(np.int64)] = oval
    return colptr, r16.astype(np.uint32), vals, bc


class ShardGroup:
    """n shards in this process; devices[i] = GPU of shard i (several shards may share one)."""

    @classmethod
    def split(cls, ctx, parts):
        """dropest_ctx_split: the reads of `ctx` (one context whose key did not fit 64 bits) as `parts` shards on its device."""
        g = cls.__new__(cls)
        g.L = capi.lib()
        g._keep = ctx
        out = (C.c_void_p * parts)()
        rc = g.L.dropest_ctx_split(ctx.h, parts, out)
        if rc != 0:
            raise capi.DropestError(rc, g.L.dropest_last_error().decode())
        g.shards = [Shard(out[i]) for i in range(parts)]
        g._handles = (C.c_void_p * parts)(*[s.h.value for s in g.shards])
        return g

    def __init__(self, devices, **cfg_kw):
        self.L = capi.lib()
        cfg, self._keep = capi.make_cfg(**cfg_kw)
        n = len(devices)
        dev = (C.c_int32 * n)(*devices)
        out = (C.c_void_p * n)()
        rc = self.L.dropest_shard_group_create(C.byref(cfg), n, dev, out)
        if rc != 0:
            raise capi.DropestError(rc, self.L.dropest_last_error().decode())
        self.shards = [Shard(out[i]) for i in range(n)]
        self._handles = (C.c_void_p * n)(*[s.h.value for s in self.shards])

    def step(self):
        rc = self.L.dropest_shard_group_step(self._handles, len(self.shards))
        if rc != 0:
            raise capi.DropestError(rc, self.L.dropest_last_error().decode())

    def close(self):
        for s in self.shards:
            s.close()


class ShardedRun:
    """One process per GPU: rank `rank` of `world`, its GPU = local_rank.  The stream's ordinal range
    [rank * reads_per_gpu, (rank + 1) * reads_per_gpu) is generated on the device once and stays resident."""

    def __init__(self, stream, rank, world, local_rank, reads_per_gpu, cfg, dist=None, **extra_cfg):
        import torch
        self.rank, self.world = rank, world
        L = capi.lib()
        uid = np.zeros(128, np.uint8)
        if rank == 0:
            rc = L.dropest_shard_unique_id(uid.ctypes.data)
            if rc != 0:
                raise capi.DropestError(rc, L.dropest_last_error().decode())
        if world > 1:
            t = torch.from_numpy(uid)
            if dist.get_backend() != "gloo":
                t = t.to(torch.device("cuda", local_rank))
            dist.broadcast(t, src=0)
            uid = t.cpu().numpy().copy()
        c, self._keep = capi.make_cfg(device=local_rank, **dict(cfg_kwargs(cfg), **extra_cfg))
        h = C.c_void_p()
        rc = L.dropest_shard_create(C.byref(c), rank, world, uid.ctypes.data, C.byref(h))
        if rc != 0:
            raise capi.DropestError(rc, L.dropest_last_error().decode())
        self.shard = Shard(h.value)
        self.R = int(reads_per_gpu)
        self.shard.set_reads(stream.generate_device(local_rank, first=rank * self.R, n=self.R), rank * self.R)

    def close(self):
        reads = self.shard._reads
        self.shard.close()
        if reads is not None:
            reads.free()
            self.shard._reads = None

    @property
    def merge_pairs(self):
        """(source, target) barcodes of the cells the CB merge folded in the last step."""
        return self.shard.merged_barcodes()

    @property
    def ctx(self):
        return self.shard.ctx

    def step(self):
        """One pass; on rank 0 returns (cm, cm_raw, cm column barcodes) with cm = (colptr, rowidx, values, column barcodes)."""
        self.shard.step()
        if self.rank != 0:
            return None, None, None
        # the form the step wrote: the 32-bit slots (the default: widened from the byte form inside the step), the byte form alone, 16-bit,
        # or 32-bit arrays -- no widening on the host here
        def take(filtered):
            form = self.shard.matrix_form(filtered)
            if form == 2:
                return self.shard.matrix_bytes(filtered)
            if form == 1:
                return self.shard.matrix_narrow(filtered)
            return self.shard.matrix(filtered)
        cm, raw = take(True), take(False)
        return cm, raw, (cm.col_barcodes if isinstance(cm, ShardBytes) else cm[3])

    def set_profiling(self, on, only=None):
        self.shard.ctx.set_profiling(on, only=only)

    def kernel_stats(self):
        return self.shard.ctx.kernel_stats()
