"""First contact of the command the driver types for N > 1 GPUs -- on a box with ONE GPU.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...

Two ranks cannot share a device under RCCL, so the launch is made with `--backend gloo --one-device` and DROPEST_SHARD_DATAPLANE=shm
(the device data of the all-to-all crosses through POSIX shared memory, csrc/shard_run.h; everything else -- the launcher, the rendezvous,
the argument handling, ShardedRun, the host collectives, the node-shared dgCMatrix slots, the timing fences and the JSON line -- is what an
8-GPU node will run).  Not a measurement: a guarantee that the command parses, launches, exchanges reads and prints a line whose matrices
are those of the N = 1 run of the same stream."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _line(res):
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]        # stdout carries exactly one JSON line
    return json.loads(lines[0])


def test_two_ranks_on_one_gpu_print_the_line_of_one_rank(tmp_path):
    common = ["--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--push-sample", "0", "--no-secondary"]
    env = dict(os.environ, PYTHONPATH=ROOT, DROPEST_BENCH_NO_BAM="1", DROPEST_BENCH_NO_FORMS="1")
    one = tmp_path / "one.npz"
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--reads", "2000000", "--cells", "400", "--dump-matrices", str(one)] + common,
                        capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    l1 = _line(r1)
    two = tmp_path / "two.npz"
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--reads", "1000000", "--cells", "200",
                         "--backend", "gloo", "--one-device", "--dump-matrices", str(two)] + common,
                        capture_output=True, text=True, env=dict(env, DROPEST_SHARD_DATAPLANE="shm"), timeout=900, cwd=ROOT)
    l2 = _line(r2)
    assert l1["n_gpus"] == 1 and l2["n_gpus"] == 2
    assert l2["steps"] == 2 and l2["warmup"] == 1 and l2["scaling"] == "weak" and l2["unit"] == "Mreads/s"
    assert l2["config"]["reads_total"] == l1["config"]["reads_total"] == 2_000_000
    assert l2["value"] > 0 and abs(l2["value"] - 2.0 / l2["ms_per_step"] * 1e3) < 0.02 * l2["value"]    # whole-job reads over the max-over-ranks time
    assert l2["config"]["matrix_form"] == l1["config"]["matrix_form"]                                   # both steps end at the dgCMatrix slots
    assert l2["exchange"] and l2["exchange"]["bytes_out_per_gpu_per_step"] > 0                          # reads really crossed between the ranks
    assert l2["roofline"] and l2["roofline"]["time_weighted_frac"] > 0
    a, b = np.load(one), np.load(two)
    for name in ("cm", "cm_raw"):
        for part in ("colptr", "rows", "vals"):
            assert np.array_equal(a[name + "_" + part], b[name + "_" + part]), (name, part)
    assert l2["config"]["cm_nnz"] == l1["config"]["cm_nnz"] and l2["config"]["cm_raw_nnz"] == l1["config"]["cm_raw_nnz"]
    assert l2["config"]["filtered_cells"] == l1["config"]["filtered_cells"] > 0
