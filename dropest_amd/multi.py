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

With cfg["merge"] (the reference's -m with a barcode whitelist) a barcode's merge target can live on another shard.
The key fields are first made identical on every shard (all-reduced ingest summary), then the merge runs in phases
(search / export / intersect / decide / apply / finish, see include/dropest_amd.h and csrc/merge_shard.h) with small
all-gathers between them: the real cells' rows, the (base, candidate) pairs, the molecule rows of the non-whitelist
bases (a few % of all molecules), the intersection sizes and the targets.  Not supported in sharded runs: barcodes or
UMIs with N (the reference's random UMI fill draws from one global rand() sequence).
"""
import ctypes as C
import os

import numpy as np

from . import capi


class GpuEngine:
    """Local compute on one MI355X through the C-ABI.  Tensors are torch CUDA tensors on `device`."""

    def __init__(self, device, cfg):
        import torch
        self.torch = torch
        self.device = device
        self.L = capi.lib()
        m = cfg.get("merge")
        kw = dict(merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=m["barcodes_kind"], barcodes_file=m["barcodes_file"],
                  min_merge_fraction=m.get("min_merge_fraction", 0.2)) if m else dict(merge_kind=capi.MERGE_NONE)
        self.ctx = capi.Context(device=device, min_genes_before_merge=cfg["min_before"],
                                min_genes_after_merge=cfg["min_after"], gene_match_levels=cfg.get("levels", "eEBA"), **kw)
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
        need = C.c_uint64()
        self.L.dropest_partition_scratch_bytes(n, C.byref(need))
        scratch = self.empty(need.value, t.uint8)           # torch's caching allocator: no hipMalloc per step
        t.cuda.synchronize(self.dev)
        rc = self.L.dropest_partition_by_owner(self.device, *[x.data_ptr() for x in reads], n, n_parts,
                                               *[x.data_ptr() for x in out], counts.ctypes.data, scratch.data_ptr(),
                                               need.value)
        if rc != 0:
            raise capi.DropestError(rc, self.L.dropest_last_error().decode())
        return out, [int(c) for c in counts]

    # ---- the single-GPU path on the owned reads, in the three pieces a sharded run needs ----
    def ingest(self, reads):
        """Barcode table + cell ids; returns the key statistics [umi_clean_min, umi_clean_max, umi_escape_max_plus1,
        gene_max_plus1, chr_max_plus1, gene_chr_conflict] (uint64)."""
        self.torch.cuda.synchronize(self.dev)
        self.ctx.clear_reads()
        self._held = reads            # adopted in place: keep the tensors alive until the next clear
        n = reads[0].numel()
        if n:
            self.ctx.push_reads_device(*[x.data_ptr() for x in reads[:4]], n, adopt=True)
        self.ctx.ingest()
        s = self.ctx.ingest_summary()
        return np.array([s.umi_clean_min, s.umi_clean_max, s.umi_escape_max_plus1, s.gene_max_plus1, s.chr_max_plus1,
                         s.gene_chr_conflict], np.uint64)

    def set_ingest_summary(self, a):
        s = capi.IngestSummary(int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5]), 0)
        self.ctx.set_ingest_summary(s)

    def gene_chr(self, n):
        """Copy of the first n entries of the gene -> chromosome table (int32, -1 = unset)."""
        p, cap = self.ctx.gene_chr_table()
        out = self.empty(min(n, cap), self.torch.int32)
        self._copy(out.data_ptr(), p, out.numel() * 4)
        return out

    def set_gene_chr(self, tensor):
        p, cap = self.ctx.gene_chr_table()
        self.torch.cuda.synchronize(self.dev)
        self._copy(p, tensor.data_ptr(), tensor.numel() * 4)

    def _copy(self, dst, src, nbytes):
        if nbytes:
            rc = self.L.dropest_dev_copy_device(self.device, dst, src, nbytes)
            if rc != 0:
                raise capi.DropestError(rc, self.L.dropest_last_error().decode())

    def initialize(self):
        self.ctx.set_initialized()
        return self.ctx.real_candidate_rows()

    def finalize(self):
        self.ctx.merge_and_filter()
        return self.ctx.real_candidate_rows()

    # ---- whitelist CB merge across shards: local phases (csrc/merge_shard.h) ----
    def merge_search(self, g_barcode, g_n_genes, g_total_umis, base_global, base_local):
        return self.ctx.shard_merge_search(g_barcode, g_n_genes, g_total_umis, base_global, base_local)

    def merge_export(self):
        t = self.torch
        listed, off, p_low, p_cols = self.ctx.shard_merge_export()
        n = int(off[-1])
        low = self.empty(n, t.int64); cols = [self.empty(n, t.int32) for _ in range(4)]
        self._copy(low.data_ptr(), p_low, n * 8)
        for c, p in zip(cols, p_cols):
            self._copy(c.data_ptr(), p, n * 4)
        return listed, off, low, cols

    def merge_intersect(self, cand_local, base_begin, base_end, low_all):
        self.torch.cuda.synchronize(self.dev)
        return self.ctx.shard_merge_intersect(cand_local, base_begin, base_end, low_all.data_ptr())

    def merge_decide(self, inter, n_bases):
        return self.ctx.shard_merge_decide(inter, n_bases)

    def merge_finish(self, local_id, excluded, merged_away, total_reads, total_umis, move_src, move_tgt, import_rows,
                     import_cell, low_all, cols_all):
        t = self.torch
        idx = t.as_tensor(np.asarray(import_rows, np.int64), device=self.dev)
        cell = t.as_tensor(np.asarray(import_cell, np.int64), device=self.dev).to(t.int32)
        low = low_all[idx] if len(import_rows) else self.empty(0, t.int64)
        cols = [c[idx] if len(import_rows) else self.empty(0, t.int32) for c in cols_all]
        t.cuda.synchronize(self.dev)
        self.ctx.shard_merge_finish(local_id, excluded, merged_away, total_reads, total_umis, move_src, move_tgt, len(import_rows),
                                    cell.data_ptr(), low.data_ptr(), [c.data_ptr() for c in cols])

    def matrix(self, filtered, as_tensors=True):
        """Local CSC pieces: colptr (numpy) + (rowidx, values) as tensors, or as the context's own device pointers
        (valid until the next matrix of the same kind)."""
        t = self.torch
        colptr, d_rows, d_vals, nnz = self.ctx.count_matrix_device(filtered=filtered)
        if not as_tensors:
            return colptr, d_rows, d_vals
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

    def register_shared(self, buf):
        addr = buf["host"].ctypes.data
        d = C.c_void_p()
        rc = self.L.dropest_host_register(self.device, addr, buf["cap"] * 8, C.byref(d))
        if rc != 0:
            raise capi.DropestError(rc, self.L.dropest_last_error().decode())
        buf["addr"], buf["dptr"] = addr, d.value

    def unregister_shared(self, buf):
        self.L.dropest_host_unregister(self.device, buf["addr"])

    deferred_writes = True      # write_columns only queues the copy; wait_writes() completes it

    def write_columns(self, src_start, dst_start, length, src_rows, src_vals, buf, slot=0):
        """This rank's columns -> their places in the shared host buffer (a kernel writing mapped host memory).  Queued
        behind the emission kernels on the device; the host goes on preparing the next matrix meanwhile."""
        s = np.ascontiguousarray(src_start, np.uint64); d = np.ascontiguousarray(dst_start, np.uint64)
        ln = np.ascontiguousarray(length, np.uint64)
        ptr = lambda x: x if isinstance(x, int) or x is None else x.data_ptr()      # noqa: E731
        # the context's kernels run on its own (non-blocking) stream, the copy on the default stream: order them.  This
        # also waits for the previous matrix's copy -- after the host work it was meant to hide.
        rc = self.L.dropest_dev_sync(self.device)
        if rc == 0:
            rc = self.L.dropest_assemble_columns_async(self.device, slot, len(s), s.ctypes.data, d.ctypes.data, ln.ctypes.data,
                                                       ptr(src_rows), ptr(src_vals), buf["dptr"], buf["dptr"] + buf["cap"] * 4)
        if rc != 0:
            raise capi.DropestError(rc, self.L.dropest_last_error().decode())

    def wait_writes(self):
        rc = self.L.dropest_dev_sync(self.device)
        if rc != 0:
            raise capi.DropestError(rc, self.L.dropest_last_error().decode())

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

    def set_profiling(self, on, only=None):
        self.ctx.set_profiling(on, only=only)


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

    def barrier(self):
        self.dist.barrier()

    def all_gather_v(self, tensor, counts):
        """Concatenation of every rank's 1-D tensor (counts[r] elements from rank r), on every rank."""
        t = self.torch
        src = self._stage(tensor)
        kmax = max(list(counts) + [1])
        pad = t.zeros(kmax, dtype=src.dtype, device=src.device)
        pad[:src.numel()] = src
        bufs = [t.empty_like(pad) for _ in range(self.world)]
        self.dist.all_gather(bufs, pad)
        out = t.cat([b[:k] for b, k in zip(bufs, counts)])
        return out.to(tensor.device) if out.device != tensor.device else out

    def all_reduce(self, tensor, op):
        """In-place all-reduce of a device tensor ("min" / "max")."""
        src = self._stage(tensor)
        self.dist.all_reduce(src, op=self.dist.ReduceOp.MIN if op == "min" else self.dist.ReduceOp.MAX)
        if src is not tensor:
            tensor.copy_(src)
        return tensor

    def all_gather_rows(self, array, as_tensors=False):
        """numpy (k, w) int64 per rank -> list of arrays (every rank gets all); as_tensors: torch tensors left on the
        collective's device (the global cell table has 10^5..10^6 rows: it is ordered there, not in numpy)."""
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
        if as_tensors:
            return [b[:n] for b, n in zip(bufs, ks)]
        return [b.cpu().numpy()[:n] for b, n in zip(bufs, ks)]


def order_cells(rows):
    """CellsDataContainer::compare_cells (CellsDataContainer.cpp:329-344) over rows with columns
    [requested_genes, requested_umis, total_umis, barcode code]: ascending; clean equal-length codes compare like the
    barcode strings."""
    return np.lexsort((rows[:, 3], rows[:, 2], rows[:, 1], rows[:, 0]))


def order_cells_device(torch, device, rows):
    """order_cells as four stable device sorts (least significant key first): numpy's lexsort takes 12 ms for the 4e5
    real cells of one C4 shard and grows with the number of ranks -- every rank orders the cells of ALL ranks."""
    t = torch.from_numpy(np.ascontiguousarray(rows)).to(device)
    order = torch.arange(t.shape[0], dtype=torch.int64, device=device)
    for col in (3, 2, 1, 0):
        order = order[torch.argsort(t[order, col], stable=True)]
    return order.cpu().numpy()


class ShardedRun:
    def __init__(self, stream, rank, world, local_rank, reads_per_gpu, cfg, dist, engine=None, staging=None):
        self.rank, self.world, self.R, self.cfg = rank, world, int(reads_per_gpu), cfg
        self.engine = engine or GpuEngine(local_rank, cfg)
        self.coll = Collectives(dist, rank, world, staging)
        self.resident = self.engine.generate(stream, rank * self.R, self.R)   # this rank's ordinal range, in HBM
        self.trace = None           # set to {} to accumulate per-phase wall times (ms)
        self.merge_pairs = None
        # where the final matrices are assembled: "shm" = every rank writes its columns into host memory shared by
        # the node's ranks; "gather" = RCCL gather onto rank 0's GPU, then one D2H
        self.output = cfg.get("output") or os.environ.get("DROPEST_SHARD_OUTPUT", "shm")
        if self.output not in ("shm", "gather"):
            raise ValueError("output must be 'shm' or 'gather'")
        self._shm, self._shm_gen = {}, 0
        tok = self.coll.all_gather_rows(np.array([[os.getpid()]], np.int64))
        self._token = int(tok[0][0, 0])

    def set_profiling(self, on, only=None):
        self.engine.set_profiling(on, only=only)

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
        # 3. local pipeline on the owned reads; the shards agree on the key fields before the keys are built
        summary = e.ingest(recv)
        if n > 1:
            self._agree_on_key_fields(summary)
        t = self._tick("ingest", t)
        ids, rows = e.initialize()
        t = self._tick("pipeline", t)
        merge_pairs = None
        if self.cfg.get("merge"):
            merge_pairs = self._cb_merge(ids, rows)
            t = self._tick("cb_merge", t)
        ids, rows = e.finalize()
        t = self._tick("finalize", t)
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
        everyone = c.all_gather_rows(table, as_tensors=True)
        t = self._tick("cells_allgather", t)
        # 5. local matrices, gathered on rank 0
        out = {}
        self._pending_writes = False
        for name, filtered in (("cm", True), ("cm_raw", False)):
            colptr, rows_t, vals_t = e.matrix(filtered, as_tensors=self.output != "shm")
            t = self._tick("emit:" + name, t)
            local_cols = e.filtered_ids().astype(np.int64) if filtered else table[:, 5]
            out[name] = self._gather_matrix(everyone, filtered, colptr, rows_t, vals_t, local_cols)
            t = self._tick("matrix:" + name, t)
        if self._pending_writes:           # both matrices are on their way to the shared host buffer: wait once, for everybody
            e.wait_writes()
            c.barrier()
            t = self._tick("matrix:wait", t)
        self.merge_pairs = merge_pairs     # (source barcode, target barcode) of every merged cell, ascending source
        return out["cm"], out["cm_raw"], out["cm"][3] if self.rank == 0 else None

    def _agree_on_key_fields(self, summary):
        """All shards must lay out the gene / UMI fields of the sort key identically (molecule rows move between
        shards in a merge) and agree on whether a gene determines its chromosome."""
        e, c = self.engine, self.coll
        rows = np.concatenate(c.all_gather_rows(summary.view(np.int64).reshape(1, 6))).view(np.uint64)
        g = np.array([rows[:, 0].min(), rows[:, 1].max(), rows[:, 2].max(), rows[:, 3].max(), rows[:, 4].max(),
                      rows[:, 5].max()], np.uint64)
        n_genes = int(g[3])
        if n_genes and not g[5]:
            tmax = e.gene_chr(n_genes)
            tmin = tmax.clone()
            tmin[tmin < 0] = 0x7FFFFFFF
            c.all_reduce(tmax, "max"); c.all_reduce(tmin, "min")
            if bool(((tmax >= 0) & (tmin != tmax)).any()):
                g[5] = 1                               # one gene on two chromosomes, seen by different shards
            else:
                e.set_gene_chr(tmax)
        e.set_ingest_summary(g)

    def _cb_merge(self, ids, rows):
        """RealBarcodes CB merge over all shards (MergeStrategyBase::merge_inited, MergeStrategyBase.cpp:11-57)."""
        e, c, rank = self.engine, self.coll, self.rank
        import time
        t_ = time.perf_counter()
        is_real = rows["is_real"].astype(bool)
        r = rows[is_real]
        if np.any(r["barcode"] >> np.uint64(63)):
            raise capi.DropestError(4, "escaped barcodes are not supported in sharded runs yet")
        # columns: 0 barcode 1 n_genes 2 total_umis 3 total_reads 4 requested_genes 5 requested_umis 6 local id
        local = np.stack([r["barcode"].astype(np.int64), r["n_genes"].astype(np.int64), r["total_umis"].astype(np.int64),
                          r["total_reads"].astype(np.int64), r["requested_genes"].astype(np.int64),
                          r["requested_umis"].astype(np.int64), ids[is_real].astype(np.int64)], axis=1)
        t_ = self._tick("cbm:local", t_)
        per_rank = c.all_gather_rows(local)
        t_ = self._tick("cbm:gather_cells", t_)
        goff = np.concatenate([[0], np.cumsum([len(x) for x in per_rank])]).astype(np.int64)
        G = np.concatenate(per_rank)
        nG, lo, hi = len(G), int(goff[rank]), int(goff[rank + 1])
        # search: my real cells against everybody's
        pb, pc = e.merge_search(G[:, 0].astype(np.uint64), G[:, 1], G[:, 2], np.arange(lo, hi), G[lo:hi, 6])
        t_ = self._tick("cbm:search", t_)
        listed, off, low_t, cols_t = e.merge_export()
        t_ = self._tick("cbm:export", t_)
        pairs = c.all_gather_rows(np.stack([pb, pc], axis=1).astype(np.int64))
        lists = c.all_gather_rows(np.stack([listed, off[:-1], off[1:]], axis=1).astype(np.int64))
        row_counts = [int(x[:, 2].max()) if len(x) else 0 for x in lists]
        row_base = np.concatenate([[0], np.cumsum(row_counts)]).astype(np.int64)
        low_all = c.all_gather_v(low_t, row_counts)
        cols_all = [c.all_gather_v(x, row_counts) for x in cols_t]
        beg = np.full(nG, -1, np.int64); end = np.full(nG, -1, np.int64)
        for q, x in enumerate(lists):
            if len(x):
                beg[x[:, 0]] = x[:, 1] + row_base[q]; end[x[:, 0]] = x[:, 2] + row_base[q]
        t_ = self._tick("cbm:gather_lists", t_)
        # intersect: the pairs whose candidate is mine
        allp = np.concatenate(pairs) if nG else np.zeros((0, 2), np.int64)
        poff = np.concatenate([[0], np.cumsum([len(x) for x in pairs])]).astype(np.int64)
        mine = np.flatnonzero((allp[:, 1] >= lo) & (allp[:, 1] < hi))
        inter = e.merge_intersect(G[allp[mine, 1], 6], beg[allp[mine, 0]], end[allp[mine, 0]], low_all)
        t_ = self._tick("cbm:intersect", t_)
        answers = c.all_gather_rows(np.stack([mine, inter.astype(np.int64)], axis=1))
        inter_all = np.zeros(len(allp), np.int64)
        for x in answers:
            if len(x):
                inter_all[x[:, 0]] = x[:, 1]
        # decide: targets of my bases; then the same sequential application everywhere
        tgt = e.merge_decide(inter_all[poff[rank]:poff[rank + 1]], hi - lo)
        target = np.concatenate(c.all_gather_rows(tgt.reshape(-1, 1)))[:, 0] if nG else np.zeros(0, np.int64)
        t_ = self._tick("cbm:decide", t_)
        keys = G[:, [4, 5, 2, 0]]                             # all real cells are "filtered" before the merge (threshold 0)
        order = order_cells_device(e.torch, e.dev, keys) if hasattr(e, "torch") else order_cells(keys)
        final, excl, reads, umis = capi.merge_apply(order, target[order], G[:, 3], G[:, 2])
        final = final.astype(np.int64)
        t_ = self._tick("cbm:order+apply", t_)
        me = np.arange(lo, hi)
        moved = np.flatnonzero(final != np.arange(nG))
        local_moves = moved[(moved >= lo) & (moved < hi) & (final[moved] >= lo) & (final[moved] < hi)]
        incoming = moved[((moved < lo) | (moved >= hi)) & (final[moved] >= lo) & (final[moved] < hi)]
        if np.any(beg[incoming] < 0):
            raise capi.DropestError(5, "internal: a merged cell's molecule rows were not exported")
        lens = end[incoming] - beg[incoming]
        import_rows = (np.concatenate([np.arange(b, b2) for b, b2 in zip(beg[incoming], end[incoming])])
                       if len(incoming) else np.zeros(0, np.int64))
        import_cell = np.repeat(G[final[incoming], 6], lens) if len(incoming) else np.zeros(0, np.int64)
        t_ = self._tick("cbm:moves", t_)
        e.merge_finish(G[me, 6], excl[me], (final[me] != me).astype(np.uint8), reads[me], umis[me], G[local_moves, 6],
                       G[final[local_moves], 6], import_rows, import_cell, low_all, cols_all)
        t_ = self._tick("cbm:finish", t_)
        return G[moved, 0].astype(np.uint64), G[final[moved], 0].astype(np.uint64)

    def _global_columns(self, everyone, metas, filtered):
        """Global column order of a matrix (identical on every rank): the kept cells' barcodes, and per column its owner
        rank, its start inside that rank's local arrays and its length -- torch tensors on the collectives' device.  The
        table has 10^5..10^6 rows at BASELINE sizes and every rank orders it every pass: a few device sorts, where
        numpy took 50-100 ms at 8 ranks."""
        import torch as t
        # table columns: [req_genes, req_umis, total_umis, barcode, first_global, local_id, n_genes]; metas: [local_id, length]
        dev = everyone[0].device
        sizes = [int(x.shape[0]) for x in everyone]
        empty = t.zeros(0, dtype=t.int64, device=dev)
        if not sum(sizes) or not any(int(m.shape[0]) for m in metas):
            return empty, empty, empty, empty
        table = t.cat(everyone)
        rank_col = t.repeat_interleave(t.arange(len(everyone), dtype=t.int64, device=dev), t.tensor(sizes, dtype=t.int64, device=dev))
        row_col = t.cat([t.arange(k, dtype=t.int64, device=dev) for k in sizes])
        if filtered:
            sel = t.nonzero(table[:, 0] >= self.cfg["min_after"]).flatten()
            # CellsDataContainer::compare_cells: lexicographic (requested_genes, requested_umis, total_umis, barcode) = stable
            # sorts from the least significant key up
            for j in (3, 2, 1, 0):
                sel = sel[t.argsort(table[sel, j], stable=True)]
        else:
            sel = t.argsort(table[:, 4], stable=True)                    # cell-id order == first-seen order
        ranks, rows = rank_col[sel], row_col[sel]
        barcodes = table[sel, 3]
        m_sizes = [int(m.shape[0]) for m in metas]
        base = t.tensor(np.concatenate([[0], np.cumsum(m_sizes)]).astype(np.int64), device=dev)
        lens_all = t.cat([m[:, 1] for m in metas])
        starts_all = t.cat([t.cumsum(m[:, 1], 0) - m[:, 1] for m in metas])
        direct = (not filtered) and m_sizes == sizes and all(k == 0 or bool(t.equal(m[:, 0], x[:, 5])) for m, x, k in zip(metas, everyone, sizes))
        if direct:
            pos = base[ranks] + rows                                     # cm_raw: a rank's columns ARE its real cells in table order
        else:
            keys = t.cat([(r << 40) | m[:, 0] for r, m in enumerate(metas)])
            o = t.argsort(keys, stable=True)
            pos = o[t.searchsorted(keys[o], (ranks << 40) | table[sel, 5])]
        return barcodes, ranks, starts_all[pos], lens_all[pos]

    def _shared(self, slot, total):
        """Host buffer of one matrix, shared by the ranks of the node: a /dev/shm file mapped (and registered with the
        GPU) by every rank, grown collectively; [rows | vals] uint32 halves of `cap` entries."""
        import mmap
        cur = self._shm.get(slot)
        if cur is not None and cur["cap"] >= total:
            return cur
        c, e = self.coll, self.engine
        if cur is not None:
            e.unregister_shared(cur)
            cur["mm"] = None
        cap = int(total * 1.25) + 4096
        self._shm_gen += 1
        path = "/dev/shm/dropest_%d_%d_%d" % (self._token, slot, self._shm_gen)
        made = 1
        if self.rank == 0:
            try:
                with open(path, "w+b") as f:
                    os.posix_fallocate(f.fileno(), 0, cap * 8)     # reserve the pages now: a full tmpfs must fail here,
            except OSError as err:                                  # not with SIGBUS in the middle of a write
                made, self._shm_error = 0, "cannot reserve %d bytes in /dev/shm: %s" % (cap * 8, err)
                try:
                    os.unlink(path)
                except OSError:
                    pass
        if not int(c.all_gather_rows(np.array([[made]], np.int64))[0][0, 0]):
            return None
        with open(path, "r+b") as f:
            mm = mmap.mmap(f.fileno(), cap * 8)
        c.barrier()
        if self.rank == 0:
            os.unlink(path)                     # the mappings keep it alive; nothing is left behind on a crash
        buf = {"mm": mm, "cap": cap, "host": np.frombuffer(mm, np.uint32)}
        try:
            e.register_shared(buf)
            ok = 1
        except capi.DropestError as err:          # e.g. a driver that cannot pin tmpfs pages
            ok, self._shm_error = 0, str(err)
        if not all(int(x[0, 0]) for x in c.all_gather_rows(np.array([[ok]], np.int64))):
            if ok:
                e.unregister_shared(buf)
            return None
        self._shm[slot] = buf
        return buf

    def _gather_matrix(self, everyone, filtered, colptr, rows_t, vals_t, local_cols):
        import time
        import torch as t
        e, c, n = self.engine, self.coll, self.world
        nnz_local = int(colptr[-1]) if len(colptr) else 0
        lens = np.diff(colptr.astype(np.int64)) if len(colptr) > 1 else np.zeros(0, np.int64)
        # tell everybody which cell each local column is and how long it is
        meta = np.stack([local_cols, lens], axis=1) if len(lens) else np.zeros((0, 2), np.int64)
        tt = time.perf_counter()
        metas = c.all_gather_rows(meta, as_tensors=True)
        tt = self._tick("gm:meta_allgather", tt)
        barcodes, col_rank, src, ln = self._global_columns(everyone, metas, filtered)
        csum = t.cumsum(ln, 0)
        dst = csum - ln
        total = int(csum[-1]) if len(ln) else 0
        host = lambda x: x.cpu().numpy()      # noqa: E731
        tt = self._tick("gm:order", tt)
        slot = 0 if filtered else 1

        def result(rows_h, vals_h):           # rank 0 only: the global CSC pieces on the host
            colptr_g = np.concatenate([[0], host(csum)]).astype(np.uint64)
            return (colptr_g, rows_h, vals_h, host(barcodes).astype(np.uint64))
        if self.output == "shm" and total > 0:
            # every rank writes ITS columns of the global matrix into the node's shared host buffer: all PCIe links
            # work at once and no GPU has to hold (or copy out) the whole matrix
            buf = self._shared(slot, total)
            if buf is None:
                if self.rank == 0:
                    import sys
                    print("dropest_amd: shared host buffer unavailable (%s); gathering over RCCL instead"
                          % getattr(self, "_shm_error", "another rank failed"), file=sys.stderr)
                self.output = "gather"
                _, rows_t, vals_t = e.matrix(filtered)       # the gather needs tensors, not the context's own arrays
            else:
                mine = col_rank == self.rank
                args = (host(src[mine]), host(dst[mine]), host(ln[mine]), rows_t, vals_t, buf)
                if getattr(e, "deferred_writes", False):
                    e.write_columns(*args, slot)
                    self._pending_writes = True
                else:
                    e.write_columns(*args)
                    c.barrier()
                tt = self._tick("gm:write_shared", tt)
                if self.rank != 0:
                    return None
                return result(buf["host"][:total], buf["host"][buf["cap"]:buf["cap"] + total])
        # "gather": all columns to rank 0's GPU over RCCL (all-to-all(v) with a single receiver), one copy kernel
        # puts them in the global order, one D2H
        nnz_all = [int(m[:, 1].sum()) for m in metas]
        send = [nnz_local if p == 0 else 0 for p in range(n)]
        recv = nnz_all if self.rank == 0 else [0] * n
        g_rows = c.all_to_all_v(rows_t, send, recv)
        g_vals = c.all_to_all_v(vals_t, send, recv)
        tt = self._tick("gm:gather", tt)
        if self.rank != 0:
            return None
        base = np.concatenate([[0], np.cumsum(nnz_all)]).astype(np.int64)
        a_rows, a_vals = e.assemble(host(src) + base[host(col_rank)], host(dst), host(ln), g_rows, g_vals, total)
        tt = self._tick("gm:assemble", tt)
        res = result(e.to_numpy_u32(a_rows, 2 * slot), e.to_numpy_u32(a_vals, 2 * slot + 1))
        self._tick("gm:d2h", tt)
        return res
