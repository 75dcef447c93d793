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


def _late_worker(token, rank, world, q, delay, rounds):
    import time
    os.environ["DROPEST_MAILBOX_TIMEOUT_S"] = "2"
    time.sleep(delay)
    _worker(token, rank, world, rounds, 64, q)


def test_a_rank_that_never_arrives_fails_the_collective_on_the_others():
    """World 3, rank 2 never starts: ranks 0 and 1 must give up after the time-out (2 s here, 300 s in a run) with an error -- not hang --,
    and nothing may be left in /dev/shm by the rank that created the object."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    token = (os.getpid() << 20) ^ 0x5151
    procs = [ctx.Process(target=_late_worker, args=(token, r, 3, q, 0.0, 10)) for r in (0, 1)]
    for p in procs:
        p.start()
    got = [q.get(timeout=60) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(rc != 0 for _, rc, _ in got), got
    assert all("did not arrive" in msg or "did not create" in msg or "failed" in msg for _, _, msg in got), got
    # rank 0 normally unlinks once everybody is attached; after a failed attach its constructor removes the object itself
    assert not any(n.startswith("dropest_mb_%x" % token) for n in os.listdir("/dev/shm"))


def test_a_rank_that_dies_between_collectives_fails_the_others():
    """World 2, both ranks in a long series of gathers; rank 1 is killed after half a second: rank 0's barrier must time out and report,
    not spin for ever."""
    import time
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    token = (os.getpid() << 20) ^ 0x7272
    procs = [ctx.Process(target=_late_worker, args=(token, r, 2, q, 0.0, 50_000_000)) for r in (0, 1)]
    for p in procs:
        p.start()
    time.sleep(1.5)
    procs[1].kill()
    rank, rc, msg = q.get(timeout=60)
    procs[0].join(timeout=30)
    assert rank == 0 and rc != 0 and "did not arrive" in msg, (rank, rc, msg)
