"""HostMailbox (csrc/shard_run.h): the host collectives of a multi-process run -- one POSIX shm object, a slot per rank in two halves used
in turn, one sense-reversing barrier per gather.  Real processes on the CPU (the RCCL transport itself needs one GPU per process)."""
import ctypes as C
import multiprocessing as mp
import os

import pytest

from dropest_amd import capi


def _worker(token, rank, world, rounds, nbytes, q):
    L = capi.lib()
    L.dropest_test_host_mailbox.restype = C.c_int
    L.dropest_test_host_mailbox.argtypes = [C.c_uint64, C.c_int32, C.c_int32, C.c_uint32, C.c_uint64]
    rc = L.dropest_test_host_mailbox(token, rank, world, rounds, nbytes)
    q.put((rank, rc, L.dropest_last_error().decode() if rc else ""))


@pytest.mark.parametrize("world,rounds,nbytes", [(1, 50, 64), (2, 2000, 8), (4, 600, 4096), (8, 300, 100_000), (3, 40, 1 << 20)])
def test_gathers_across_processes(world, rounds, nbytes):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    token = (os.getpid() << 20) ^ (world * 7919 + rounds)
    procs = [ctx.Process(target=_worker, args=(token, r, world, rounds, nbytes, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r for r, _, _ in got) == list(range(world))
    assert all(rc == 0 for _, rc, _ in got), got
    assert not any(n.startswith("dropest_mb_") for n in os.listdir("/dev/shm"))       # unlinked once everybody was attached
