"""Two ranks of the sharded run with the REAL engine (HIP path) on one GPU: the collectives hop through host memory
over gloo (two ranks cannot share a device under RCCL), everything else is the production code path.  The assembled
matrices must equal a single-context run over the whole stream."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from dropest_amd import capi
from dropest_amd.multi import ShardedRun
from dropest_amd.synth import SynthStream

pytestmark = pytest.mark.gpu

# soak runs: DROPEST_MULTI_SCALE=<k> multiplies reads and cells of every case of this file (default 1 = the committed cases)
SCALE = int(os.environ.get("DROPEST_MULTI_SCALE", "1"))
CFG = {"min_before": 10, "min_after": 30}
STREAM = dict(n_reads=400_000 * SCALE, n_cells=60 * SCALE, n_genes=3000)


DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dropest_amd", "data", "barcodes")
MERGE_CASES = {
    # name: (stream parameters, whitelist file, barcodes kind, thresholds)
    "10x": (dict(n_reads=300_000 * SCALE, n_cells=40 * SCALE, n_genes=2000, umi_len=12, permille_neighbour=150), "10x_aug_2016_split",
            capi.BARCODES_CONST, {"min_before": 3, "min_after": 20}),
    "indrop": (dict(n_reads=300_000 * SCALE, n_cells=40 * SCALE, n_genes=2000, umi_len=8, permille_neighbour=150, whitelist="indrop_v3"),
               "indrop_v3", capi.BARCODES_CONST, {"min_before": 3, "min_after": 20}),
}


def _worker(rank, world, port, path, case=None, output="shm"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["DROPEST_SHARD_OUTPUT"] = output
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if case is None:
            kw, cfg = STREAM, CFG
        else:
            kw, wl, kind, cfg = MERGE_CASES[case]
            cfg = dict(cfg, merge={"barcodes_kind": kind, "barcodes_file": os.path.join(DATA, wl)})
        stream = SynthStream(**kw)
        run = ShardedRun(stream, rank, world, 0, kw["n_reads"] // world, cfg, dist, staging="cpu")
        for _ in range(2):                      # a second step exercises clear_reads / buffer reuse
            cm, cm_raw, cols = run.step()
        if rank == 0:
            extra = {}
            if run.merge_pairs is not None:
                extra = dict(m_src=run.merge_pairs[0], m_tgt=run.merge_pairs[1])
            np.savez(path, cm_p=cm[0], cm_i=cm[1], cm_x=cm[2], cm_cols=cm[3], raw_p=cm_raw[0], raw_i=cm_raw[1],
                     raw_x=cm_raw[2], raw_cols=cm_raw[3], **extra)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("output", ["shm", "gather"])
def test_two_ranks_on_one_gpu_match_single_context(output, tmp_path):
    """output = "shm": every rank writes its columns into host memory shared by the ranks (registered /dev/shm
    mapping); "gather": the columns are gathered on rank 0's GPU first."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    path = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(2, port, path, None, output), nprocs=2, join=True)
    got = np.load(path)
    stream = SynthStream(**STREAM)
    dev = stream.generate_device(0)
    c = capi.Context(min_genes_before_merge=CFG["min_before"], min_genes_after_merge=CFG["min_after"])
    c.push_reads_device(*dev.ptrs, dev.n, adopt=True)
    c.set_initialized(); c.merge_and_filter()
    rows = c.cell_rows()
    for filt, pre in ((True, "cm"), (False, "raw")):
        p, i, x = c.count_matrix_csc(filtered=filt)
        assert np.array_equal(got[pre + "_p"].astype(np.uint32), p)
        assert np.array_equal(got[pre + "_i"], i) and np.array_equal(got[pre + "_x"], x)
    assert [int(b) for b in got["cm_cols"]] == [int(rows["barcode"][int(k)]) for k in c.filtered_cells()]
    assert [int(b) for b in got["raw_cols"]] == [int(b) for b in rows["barcode"][rows["is_real"].astype(bool)]]
    assert len(got["cm_cols"]) > 20
    dev.free()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", sorted(MERGE_CASES))
def test_sharded_whitelist_merge_matches_single_context(case, world, tmp_path):
    """-m with a whitelist over 2 / 3 shards: merge targets on other shards, molecule rows moving between shards.
    Reference result: ONE context over the whole stream (itself pinned on the oracle in test_gpu_parity.py)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    path = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(world, port, path, case), nprocs=world, join=True)
    got = np.load(path)
    kw, wl, kind, cfg = MERGE_CASES[case]
    stream = SynthStream(**kw)
    n = (kw["n_reads"] // world) * world
    dev = stream.generate_device(0, first=0, n=n)
    c = capi.Context(merge_kind=capi.MERGE_REAL_BARCODES, barcodes_kind=kind, barcodes_file=os.path.join(DATA, wl),
                     min_genes_before_merge=cfg["min_before"], min_genes_after_merge=cfg["min_after"])
    c.push_reads_device(*dev.ptrs, dev.n, adopt=True)
    c.set_initialized(); c.merge_and_filter()
    rows = c.cell_rows()
    for filt, pre in ((True, "cm"), (False, "raw")):
        p, i, x = c.count_matrix_csc(filtered=filt)
        assert np.array_equal(got[pre + "_p"].astype(np.uint32), p)
        assert np.array_equal(got[pre + "_i"], i) and np.array_equal(got[pre + "_x"], x)
    assert [int(b) for b in got["cm_cols"]] == [int(rows["barcode"][int(k)]) for k in c.filtered_cells()]
    assert [int(b) for b in got["raw_cols"]] == [int(b) for b in rows["barcode"][rows["is_real"].astype(bool)]]
    mt = c.merge_targets()
    src = np.flatnonzero(mt != np.arange(len(mt)))
    want = {int(rows["barcode"][k]): int(rows["barcode"][int(mt[k])]) for k in src}
    have = dict(zip((int(b) for b in got["m_src"]), (int(b) for b in got["m_tgt"])))
    assert have == want and len(want) > 20
    assert int(rows["is_excluded"].sum()) > 0
    # the sharded merge really crossed shards
    owner = lambda b: capi.lib().dropest_owner_of(int(b), world)           # noqa: E731
    assert sum(owner(a) != owner(b) for a, b in want.items()) > 5
    dev.free()
