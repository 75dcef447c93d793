"""One process per GPU: the Estimation hot path sharded by cell barcode.

The reference has no distributed runtime; this is a new design for MI355X nodes.  Every per-read step of the path is
independent per barcode, so reads are sharded by owner(cb) = mix64(cb) mod n:

  1. each rank holds a contiguous ordinal range of the stream in HBM and partitions it by owner, stably
     (dropest_partition_by_owner: one radix pass + gather);
  2. ONE all-to-all(v) over RCCL / xGMI moves every read to its owner (5 arrays, 28 B/read) -- the only data-path
     collective; received blocks arrive in source-rank order, i.e. still in global stream order;
  3. the single-GPU pipeline runs on what the rank owns (cell ids = local first-seen ranks);
  4. small collectives: real cells' (barcode, sizes, global first ordinal) are all-gathered, rank 0 orders them with the
     reference's compare_cells key and assigns matrix columns;
  5. the per-shard count matrices (CSC) are gathered on rank 0 (all-to-all(v) with a single receiver) and their
     columns are put in the global order by one copy kernel.

`torch.distributed` (backend "nccl" = RCCL) is the transport; the compute is the C-ABI library.  The engine that
does the local compute is injected, so the orchestration can be exercised on CPU tensors over gloo (tests only).
CB merge across shards (a barcode's merge target can live on another GPU) is not built yet: merge_kind must be NONE.
"""
import ctypes as C

import numpy as np

from . import capi


class GpuEngine:
    """Local compute on one MI355X through the C-ABI.  Tensors are torch CUDA tensors on `device`."""

    def __init__(self, device, cfg):
        import torch
        self.torch = torch
        self.device = device
        self.L = capi.lib()
        self.ctx = capi.Context(device=device, merge_kind=capi.MERGE_NONE, min_genes_before_merge=cfg["min_before"],
                                min_genes_after_merge=cfg["min_after"], gene_match_levels=cfg.get("levels", "eEBA"))
        self.dev = torch.device("cuda", device)

    def empty(self, n, dtype):
        return self.torch.empty(int(n), dtype=dtype, device=self.dev)

    def generate(self, stream, first, n):
        t = self.torch
        out = [self.empty(n, t.int64), self.empty(n, t.int64), self.empty(n, t.int32), self.empty(n, t.int32)]
        rc = self.L.dropest_synth_generate_device(C.byref(stream.params), self.device, first, n, *[x.data_ptr() for x in out])
        if rc != 0:
            raise RuntimeError("device generation failed (%d)" % rc)
        return out

    def partition(self, reads, n_parts):
        t = self.torch
        n = reads[0].numel()
        out = [self.empty(n, t.int64), self.empty(n, t.int64), self.empty(n, t.int32), self.empty(n, t.int32), self.empty(n, t.int32)]
        counts = np.zeros(n_parts, np.uint64)
        t.cuda.synchronize(self.dev)
        rc = self.L.dropest_partition_by_owner(self.device, *[x.data_ptr() for x in reads], n, n_parts,
                                               *[x.data_ptr() for x in out], counts.ctypes.data)
        if rc != 0:
            raise capi.DropestError(rc, self.L.dropest_last_error().decode())
        return out, [int(c) for c in counts]

    def pipeline(self, reads):
        """Runs the single-GPU path on the owned reads; returns the real-candidate cells."""
        self.torch.cuda.synchronize(self.dev)
        self.ctx.clear_reads()
        self._held = reads            # adopted in place: keep the tensors alive until the next clear
        n = reads[0].numel()
        if n:
            self.ctx.push_reads_device(*[x.data_ptr() for x in reads[:4]], n, adopt=True)
        self.ctx.set_initialized()
        self.ctx.merge_and_filter()
        ids, rows = self.ctx.real_candidate_rows()
        return ids, rows

    def matrix(self, filtered, col_ids_expected=None):
        """Local CSC pieces as tensors (rowidx, values) + colptr (numpy)."""
        t = self.torch
        colptr, d_rows, d_vals, nnz = self.ctx.count_matrix_device(filtered=filtered)
        rows = self.empty(nnz, t.int32); vals = self.empty(nnz, t.int32)
        if nnz:
            for dst, src in ((rows, d_rows), (vals, d_vals)):
                rc = self.L.dropest_dev_copy_device(self.device, dst.data_ptr(), src, nnz * 4)
                if rc != 0:
                    raise capi.DropestError(rc, self.L.dropest_last_error().decode())
        return colptr, rows, vals

    def filtered_ids(self):
        return self.ctx.filtered_cells()

    def assemble(self, src_start, dst_start, length, src_rows, src_vals, total):
        t = self.torch
        dst_rows = self.empty(total, t.int32); dst_vals = self.empty(total, t.int32)
        t.cuda.synchronize(self.dev)
        s = np.ascontiguousarray(src_start, np.uint64); d = np.ascontiguousarray(dst_start, np.uint64)
        ln = np.ascontiguousarray(length, np.uint64)
        rc = self.L.dropest_assemble_columns(self.device, len(s), s.ctypes.data, d.ctypes.data, ln.ctypes.data,
                                             src_rows.data_ptr(), src_vals.data_ptr(), dst_rows.data_ptr(), dst_vals.data_ptr())
        if rc != 0:
            raise capi.DropestError(rc, self.L.dropest_last_error().decode())
        return dst_rows, dst_vals

    def to_numpy_u32(self, tensor, slot=0):
        """Device tensor -> numpy view of a persistent PINNED host buffer (valid until the next call with the same slot):
        a pageable .cpu() copy runs at a fraction of the PCIe rate and page-faults fresh memory every step."""
        t = self.torch
        n = tensor.numel()
        if not hasattr(self, "_pinned"):
            self._pinned = {}
        buf = self._pinned.get(slot)
        if buf is None or buf.numel() < n:
            buf = t.empty(max(int(n * 1.25), 1024), dtype=t.int32, pin_memory=True)
            self._pinned[slot] = buf
        buf[:n].copy_(tensor, non_blocking=True)
        t.cuda.synchronize(self.dev)
        return buf[:n].numpy().view(np.uint32)

    def take(self, tensor, positions):
        t = self.torch
        if len(positions) == 0:
            return np.zeros(0, np.int64)
        idx = t.as_tensor(np.asarray(positions, np.int64), device=self.dev)
        return tensor[idx].cpu().numpy().astype(np.int64)

    def kernel_stats(self):
        return self.ctx.kernel_stats()

    def set_profiling(self, on):
        self.ctx.set_profiling(on)


class Collectives:
    """torch.distributed wrappers.  With staging="cpu" tensors hop through host memory (gloo): used to run two
    ranks on ONE GPU in tests; the production path hands the device tensors to RCCL directly."""

    def __init__(self, dist, rank, world, staging=None):
        import torch
        self.torch, self.dist, self.rank, self.world, self.staging = torch, dist, rank, world, staging

    def _stage(self, t):
        return t.cpu() if self.staging == "cpu" else t

    def all_to_all_counts(self, counts):
        """recv_counts[p] = what rank p sends to me (an all-gather of the send vectors: works on every backend)."""
        t = self.torch
        dev = "cpu" if self.staging == "cpu" or self.dist.get_backend() == "gloo" else "cuda"
        send = t.tensor(counts, dtype=t.int64, device=dev)
        rows = [t.empty_like(send) for _ in range(self.world)]
        self.dist.all_gather(rows, send)
        return [int(r[self.rank].item()) for r in rows]

    def all_to_all_v(self, tensor, send_counts, recv_counts):
        t = self.torch
        src = self._stage(tensor)
        out = t.empty(int(sum(recv_counts)), dtype=src.dtype, device=src.device)
        if self.dist.get_backend() == "gloo":
            # gloo has no all_to_all_single with uneven splits on every build: use point-to-point rounds
            outs = list(out.split(recv_counts)) if sum(recv_counts) else [out[:0] for _ in recv_counts]
            ins = list(src.split(send_counts)) if sum(send_counts) else [src[:0] for _ in send_counts]
            reqs = []
            for peer in range(self.world):
                if peer == self.rank:
                    outs[peer].copy_(ins[peer])
                    continue
                if send_counts[peer]:
                    reqs.append(self.dist.isend(ins[peer].contiguous(), peer))
                if recv_counts[peer]:
                    reqs.append(self.dist.irecv(outs[peer], peer))
            for r in reqs:
                r.wait()
        else:
            self.dist.all_to_all_single(out, src, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts))
        return out.to(tensor.device) if out.device != tensor.device else out

    def all_gather_rows(self, array):
        """numpy (k, w) int64 per rank -> list of arrays (every rank gets all)."""
        t = self.torch
        backend_cpu = self.staging == "cpu" or self.dist.get_backend() == "gloo"
        dev = "cpu" if backend_cpu else "cuda"
        k = t.tensor([array.shape[0]], dtype=t.int64, device=dev)
        ks = [t.zeros_like(k) for _ in range(self.world)]
        self.dist.all_gather(ks, k)
        ks = [int(x.item()) for x in ks]
        width = array.shape[1]
        kmax = max(ks + [1])
        pad = np.zeros((kmax, width), np.int64)
        pad[:array.shape[0]] = array
        mine = t.from_numpy(pad).to(dev)
        bufs = [t.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(bufs, mine)
        return [b.cpu().numpy()[:n] for b, n in zip(bufs, ks)]


def order_cells(rows):
    """CellsDataContainer::compare_cells (CellsDataContainer.cpp:329-344) over rows with columns
    [requested_genes, requested_umis, total_umis, barcode code]: ascending; clean equal-length codes compare like the
    barcode strings."""
    return np.lexsort((rows[:, 3], rows[:, 2], rows[:, 1], rows[:, 0]))


class ShardedRun:
    def __init__(self, stream, rank, world, local_rank, reads_per_gpu, cfg, dist, engine=None, staging=None):
        self.rank, self.world, self.R, self.cfg = rank, world, int(reads_per_gpu), cfg
        self.engine = engine or GpuEngine(local_rank, cfg)
        self.coll = Collectives(dist, rank, world, staging)
        self.resident = self.engine.generate(stream, rank * self.R, self.R)   # this rank's ordinal range, in HBM
        self.trace = None           # set to {} to accumulate per-phase wall times (ms)

    def set_profiling(self, on):
        self.engine.set_profiling(on)

    def kernel_stats(self):
        return self.engine.kernel_stats()

    def _tick(self, name, t0):
        import time
        if self.trace is not None:
            if hasattr(self.engine, "torch"):
                self.engine.torch.cuda.synchronize()
            self.trace[name] = self.trace.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return time.perf_counter()

    def step(self):
        import time
        e, c, n = self.engine, self.coll, self.world
        t = time.perf_counter()
        # 1-2. partition by owner, all-to-all
        parts, send_counts = e.partition(self.resident, n)
        t = self._tick("partition", t)
        recv_counts = c.all_to_all_counts(send_counts)
        recv = [c.all_to_all_v(x, send_counts, recv_counts) for x in parts]
        t = self._tick("all_to_all", t)
        # 3. local pipeline on the owned reads
        ids, rows = e.pipeline(recv)
        t = self._tick("pipeline", t)
        # 4. global view of the real cells
        offs = np.concatenate([[0], np.cumsum(recv_counts)])
        first_pos = rows["first_read"].astype(np.int64)
        src_rank = np.searchsorted(offs, first_pos, side="right") - 1
        first_global = src_rank * self.R + e.take(recv[4], first_pos)
        is_real = rows["is_real"].astype(bool)
        table = np.stack([rows["requested_genes"].astype(np.int64), rows["requested_umis"].astype(np.int64),
                          rows["total_umis"].astype(np.int64), rows["barcode"].astype(np.int64), first_global,
                          ids.astype(np.int64), rows["n_genes"].astype(np.int64)], axis=1)[is_real]
        if np.any(rows["barcode"][is_real] >> np.uint64(63)):
            raise capi.DropestError(4, "escaped barcodes are not supported in sharded runs yet")
        everyone = c.all_gather_rows(table)
        t = self._tick("cells_allgather", t)
        # 5. local matrices, gathered on rank 0
        out = {}
        for name, filtered in (("cm", True), ("cm_raw", False)):
            colptr, rows_t, vals_t = e.matrix(filtered)
            t = self._tick("emit:" + name, t)
            local_cols = e.filtered_ids().astype(np.int64) if filtered else table[:, 5]
            out[name] = self._gather_matrix(everyone, filtered, colptr, rows_t, vals_t, local_cols)
            t = self._tick("matrix:" + name, t)
        return out["cm"], out["cm_raw"], out["cm"][3] if self.rank == 0 else None

    def _gather_matrix(self, everyone, filtered, colptr, rows_t, vals_t, local_cols):
        e, c, n = self.engine, self.coll, self.world
        nnz_local = int(colptr[-1]) if len(colptr) else 0
        lens = np.diff(colptr.astype(np.int64)) if len(colptr) > 1 else np.zeros(0, np.int64)
        # tell rank 0 which cell each local column is and how long it is
        meta = np.stack([local_cols, lens], axis=1) if len(lens) else np.zeros((0, 2), np.int64)
        import time
        tt = time.perf_counter()
        metas = c.all_gather_rows(meta)
        tt = self._tick("gm:meta_allgather", tt)
        nnz_all = [int(m[:, 1].sum()) for m in metas]
        send = [nnz_local if p == 0 else 0 for p in range(n)]
        recv = nnz_all if self.rank == 0 else [0] * n
        g_rows = c.all_to_all_v(rows_t, send, recv)
        g_vals = c.all_to_all_v(vals_t, send, recv)
        tt = self._tick("gm:gather", tt)
        if self.rank != 0:
            return None
        # global column order on rank 0
        cells = np.concatenate([np.concatenate([np.full((len(t), 1), r, np.int64), t], axis=1) for r, t in enumerate(everyone)])
        # columns: [rank, req_genes, req_umis, total_umis, barcode, first_global, local_id, n_genes]
        if filtered:
            keep = cells[cells[:, 1] >= self.cfg["min_after"]]
            order = order_cells(keep[:, 1:5])
        else:
            keep = cells
            order = np.argsort(keep[:, 5], kind="stable")       # cell-id order == first-seen order
        keep = keep[order]
        # locate every kept column inside the gathered buffer: key = (rank, local cell id)
        keys, starts, lens_all = [], [], []
        base = 0
        for r, m in enumerate(metas):
            if len(m):
                keys.append((np.int64(r) << 40) | m[:, 0])
                starts.append(base + np.concatenate([[0], np.cumsum(m[:, 1])[:-1]]))
                lens_all.append(m[:, 1])
            base += nnz_all[r]
        if keys:
            keys = np.concatenate(keys); starts = np.concatenate(starts); lens_all = np.concatenate(lens_all)
            o = np.argsort(keys, kind="stable")
            pos = np.searchsorted(keys[o], (keep[:, 0] << 40) | keep[:, 6])
            src, ln = starts[o][pos], lens_all[o][pos]
        else:
            src, ln = np.zeros(0, np.int64), np.zeros(0, np.int64)
        ln = np.asarray(ln, np.int64)
        dst = np.concatenate([[0], np.cumsum(ln)[:-1]]) if len(ln) else np.zeros(0, np.int64)
        total = int(ln.sum())
        tt = self._tick("gm:order", tt)
        a_rows, a_vals = e.assemble(np.asarray(src, np.int64), dst, ln, g_rows, g_vals, total)
        tt = self._tick("gm:assemble", tt)
        colptr_g = np.concatenate([[0], np.cumsum(ln)]).astype(np.uint64)
        slot = 0 if filtered else 2
        res = (colptr_g, e.to_numpy_u32(a_rows, slot), e.to_numpy_u32(a_vals, slot + 1), keep[:, 4].astype(np.uint64))
        self._tick("gm:d2h", tt)
        return res
        return colptr_g, e.to_numpy_u32(a_rows, slot), e.to_numpy_u32(a_vals, slot + 1), keep[:, 4].astype(np.uint64)
