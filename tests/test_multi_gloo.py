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

    def set_profiling(self, on, only=None):
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


# ---------------------------------------------------------------------------------------------------------------
# whitelist CB merge across shards: the orchestration of multi.py's _cb_merge over gloo, world size 2
# ---------------------------------------------------------------------------------------------------------------
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dropest_amd", "data", "barcodes")
MCFG = {"min_before": 3, "min_after": 10,
        "merge": {"barcodes_kind": capi.BARCODES_CONST, "barcodes_file": os.path.join(DATA, "10x_aug_2016_split"),
                  "min_merge_fraction": 0.2}}
MSTREAM = dict(n_reads=80_000, n_cells=20, n_genes=700, umi_len=8, permille_neighbour=160)
QUERY = (2, 3, 6, 7)                      # -L eEBA (UMI.cpp:123-154)


class CpuMergeEngine(CpuEngine):
    """TEST-ONLY: the per-shard container is a dict of molecules taken from the CPU oracle; the merge phases are
    restated in a few lines of Python each (whitelist distances come from the oracle's BarcodesParser restatement)."""

    def _rows(self):                      # initialize() and finalize() both land here
        if getattr(self, "state", None) is None:
            return self._initialize()
        return self._finalize()

    def ingest(self, reads):
        self.state = None
        return CpuEngine.ingest(self, reads)

    def _initialize(self):
        reads = self.reads
        cb = reads[0].numpy().view(np.uint64); umi = reads[1].numpy().view(np.uint64)
        gene = reads[2].numpy().view(np.uint32); aux = reads[3].numpy().view(np.uint32)
        m = MCFG["merge"]
        o = Oracle(merge_kind=1, barcodes_kind=1, barcodes_file=m["barcodes_file"], min_genes_before=MCFG["min_before"],
                   min_genes_after=MCFG["min_after"])
        o.add_packed(cb, umi, gene, aux)
        o.set_initialized()                                    # no merge_and_filter: the merge is done across shards below
        self.o = o
        raw_of = np.array([int(o.gene_name(i)[1:]) for i in range(o.n_genes)], np.int64)
        cell, g, umis, nreads, mark = o.molecules()
        self.mol = {}
        for c, gg, u, r, mk in zip(cell, g, umis, nreads, mark):
            self.mol.setdefault(int(c), {})[(int(raw_of[int(gg)]) << 32) | capi.pack_seq(u)] = [int(r), int(mk)]
        orows = o.cell_rows()
        _, first = np.unique(cb, return_index=True)
        first = np.sort(first)
        self.barcode = cb[first]; self.first = first
        n = o.n_cells
        self.state = dict(merged=np.zeros(n, bool), excluded=np.zeros(n, bool), total_reads=orows[:, 6].copy(),
                          total_umis=orows[:, 7].copy())
        return self._finalize()

    def _finalize(self):
        n, st = self.o.n_cells, self.state
        rows = np.zeros(n, capi.CELL_ROW_DTYPE)
        rows["barcode"] = self.barcode; rows["first_read"] = self.first
        for c in range(n):
            genes, req_genes, req_umis = set(), set(), 0
            for k, (r, mk) in self.mol.get(c, {}).items():
                genes.add(k >> 32)
                if mk in QUERY:
                    req_genes.add(k >> 32); req_umis += 1
            rows["n_genes"][c] = len(genes); rows["requested_genes"][c] = len(req_genes); rows["requested_umis"][c] = req_umis
        rows["total_reads"] = st["total_reads"]; rows["total_umis"] = st["total_umis"]
        rows["is_merged"] = st["merged"]; rows["is_excluded"] = st["excluded"]
        rows["is_real"] = ~st["merged"] & ~st["excluded"] & (rows["n_genes"] >= MCFG["min_before"])
        self.rows = rows
        keep = np.nonzero(rows["is_real"] | st["merged"] | st["excluded"] | (rows["n_genes"] >= MCFG["min_before"]))[0]
        return keep.astype(np.uint64), rows[keep]

    # ---- merge phases ----
    def merge_search(self, g_barcode, g_n_genes, g_total_umis, base_global, base_local):
        index = {int(b): g for g, b in enumerate(g_barcode)}
        parts = [self.o.wl_part(0), self.o.wl_part(1)]
        self.search = dict(base_global=np.asarray(base_global), base_local=np.asarray(base_local), cands=[],
                           g_total_umis=np.asarray(g_total_umis))
        pb, pc = [], []
        for g, loc in zip(base_global, base_local):
            cbs = capi.unpack_code(int(g_barcode[g]))
            dist = []
            for p in (0, 1):
                vals, idx = self.o.wl_distances(cbs, p)
                d = np.zeros(len(parts[p]), np.int64); d[idx.astype(np.int64)] = vals
                dist.append(d)
            cands = []
            for level in range(6):                                         # RealBarcodesMergeStrategy.cpp:82-106
                for d0 in range(level + 1):
                    for i in np.nonzero(dist[0] == d0)[0]:
                        for j in np.nonzero(dist[1] == level - d0)[0]:
                            gg = index.get(capi.pack_seq(parts[0][i] + parts[1][j]))
                            if gg is not None and g_n_genes[gg] >= MCFG["min_before"] and g_total_umis[gg] >= g_total_umis[g]:
                                cands.append(gg)
                if cands:
                    break
            self.search["cands"].append(cands)
            if cands and g not in cands:
                pb += [g] * len(cands); pc += cands
        return np.array(pb, np.uint32), np.array(pc, np.uint32)

    def merge_export(self):
        s = self.search
        listed, off, low = [], [0], []
        for g, loc, cands in zip(s["base_global"], s["base_local"], s["cands"]):
            if cands and g not in cands:
                keys = sorted(self.mol.get(int(loc), {}))
                listed.append(g); low += keys; off.append(len(low))
        s["export"] = [self.mol[int(loc)] for g, loc, c in zip(s["base_global"], s["base_local"], s["cands"]) if c and g not in c]
        vals = [[v[k][0] for v in s["export"] for k in sorted(v)], [v[k][1] for v in s["export"] for k in sorted(v)]]
        z = torch.zeros(len(low), dtype=torch.int32)
        return (np.array(listed, np.uint32), np.array(off, np.uint64), torch.tensor(low, dtype=torch.int64),
                [torch.tensor(vals[0], dtype=torch.int32), torch.tensor(vals[1], dtype=torch.int32), z, z.clone()])

    def merge_intersect(self, cand_local, base_begin, base_end, low_all):
        low = low_all.numpy()
        return np.array([len(set(low[int(b):int(e)].tolist()) & set(self.mol.get(int(c), {})))
                         for c, b, e in zip(cand_local, base_begin, base_end)], np.uint32)

    def merge_decide(self, inter, n_bases):
        s, out, p = self.search, np.full(n_bases, -1, np.int64), 0
        for f, (g, cands) in enumerate(zip(s["base_global"], s["cands"])):
            if not cands:
                continue
            if g in cands:
                out[f] = g; continue
            fr = [0.5 * int(inter[p + k]) * (1. / s["g_total_umis"][g] + 1. / s["g_total_umis"][c]) for k, c in enumerate(cands)]
            p += len(cands)
            best = max(fr)
            assert fr.count(best) == 1 or best < MCFG["merge"]["min_merge_fraction"], "tie: pick another test stream"
            out[f] = cands[fr.index(best)] if best >= MCFG["merge"]["min_merge_fraction"] else -1
        return out

    def merge_finish(self, local_id, excluded, merged_away, total_reads, total_umis, move_src, move_tgt, import_rows,
                     import_cell, low_all, cols_all):
        st = self.state
        li = np.asarray(local_id, np.int64)
        st["excluded"][li] = np.asarray(excluded, bool); st["merged"][li] = np.asarray(merged_away, bool)
        st["total_reads"][li] = total_reads; st["total_umis"][li] = total_umis

        def add(tgt, key, r, mk):                               # Gene::merge: counts add, marks OR (Gene.cpp:26-36)
            cur = self.mol.setdefault(int(tgt), {}).setdefault(int(key), [0, 0])
            cur[0] += int(r); cur[1] |= int(mk)
        for s_, t_ in zip(move_src, move_tgt):
            for k, (r, mk) in self.mol.pop(int(s_), {}).items():
                add(t_, k, r, mk)
        low = low_all.numpy(); rd = cols_all[0].numpy(); mk = cols_all[1].numpy()
        for row, cell in zip(import_rows, import_cell):
            add(cell, low[int(row)], rd[int(row)], mk[int(row)])

    # ---- matrices from the molecule dict ----
    def filtered_ids(self):
        r = self.rows
        ids = np.nonzero(r["is_real"] & (r["requested_genes"] >= MCFG["min_after"]))[0]
        key = np.stack([r["requested_genes"][ids].astype(np.int64), r["requested_umis"][ids].astype(np.int64),
                        r["total_umis"][ids].astype(np.int64), r["barcode"][ids].astype(np.int64)], axis=1)
        return ids[np.lexsort((key[:, 3], key[:, 2], key[:, 1], key[:, 0]))].astype(np.uint64)

    def matrix(self, filtered, as_tensors=True):
        cols = self.filtered_ids() if filtered else np.nonzero(self.rows["is_real"])[0]
        colptr, g, v = [0], [], []
        for c in cols:
            per_gene = {}
            for k, (r, mk) in self.mol.get(int(c), {}).items():
                if not filtered or mk in QUERY:
                    per_gene[k >> 32] = per_gene.get(k >> 32, 0) + 1
            for gene in sorted(per_gene):
                g.append(gene); v.append(per_gene[gene])
            colptr.append(len(g))
        return np.array(colptr, np.uint32), torch.tensor(g, dtype=torch.int32), torch.tensor(v, dtype=torch.int32)


def _merge_worker(rank, world, port, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stream = SynthStream(**MSTREAM)
        run = ShardedRun(stream, rank, world, 0, MSTREAM["n_reads"] // world, MCFG, dist, engine=CpuMergeEngine())
        cm, cm_raw, cols = run.step()
        if rank == 0:
            np.savez(result_path, cm_p=cm[0], cm_i=cm[1], cm_x=cm[2], cm_cols=cm[3], raw_p=cm_raw[0], raw_i=cm_raw[1],
                     raw_x=cm_raw[2], raw_cols=cm_raw[3], m_src=run.merge_pairs[0], m_tgt=run.merge_pairs[1])
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_whitelist_merge_orchestration(world, tmp_path):
    """The merge phases of dropest_amd/multi.py (_cb_merge) with targets on other shards, over gloo; reference result:
    ONE oracle container with -m over the whole stream."""
    path = str(tmp_path / "res.npz")
    mp.spawn(_merge_worker, args=(world, _free_port(), path), nprocs=world, join=True)
    got = np.load(path)
    stream = SynthStream(**MSTREAM)
    n = (MSTREAM["n_reads"] // world) * world
    cb, umi, gene, aux = stream.generate_host(0, n)
    m = MCFG["merge"]
    o = Oracle(merge_kind=1, barcodes_kind=1, barcodes_file=m["barcodes_file"], min_genes_before=MCFG["min_before"],
               min_genes_after=MCFG["min_after"])
    o.add_packed(cb, umi, gene, aux)
    o.set_initialized(); o.merge_and_filter()
    raw_of = np.array([int(o.gene_name(i)[1:]) for i in range(o.n_genes)], np.int64)
    for filt, pre in ((True, "cm"), (False, "raw")):
        g, c, v = o.count_matrix(filtered=filt)
        g = raw_of[g.astype(np.int64)] if len(g) else np.zeros(0, np.int64)
        order = np.lexsort((g, c.astype(np.int64)))
        ncols = len(o.filtered_cells()) if filt else o.n_real
        colptr = np.concatenate([[0], np.cumsum(np.bincount(c.astype(np.int64), minlength=ncols))])
        assert np.array_equal(got[pre + "_p"].astype(np.int64), colptr)
        assert np.array_equal(got[pre + "_i"].astype(np.int64), g[order])
        assert np.array_equal(got[pre + "_x"].astype(np.int64), v.astype(np.int64)[order])
    assert [int(x) for x in got["cm_cols"]] == [capi.pack_seq(o.cell_barcode(int(i))) for i in o.filtered_cells()]
    mt = o.merge_targets()
    want = {capi.pack_seq(o.cell_barcode(i)): capi.pack_seq(o.cell_barcode(int(t))) for i, t in enumerate(mt) if int(t) != i}
    have = dict(zip((int(b) for b in got["m_src"]), (int(b) for b in got["m_tgt"])))
    assert have == want and len(want) > 10
    owner = lambda b: capi.lib().dropest_owner_of(int(b), world)           # noqa: E731
    assert sum(owner(a) != owner(b) for a, b in want.items()) > 2          # targets really lived on other shards
