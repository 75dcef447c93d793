"""Developer tool: per-stage wall times (device-synchronised) of the sharded step, world = 1 over RCCL."""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"
import torch, torch.distributed as dist
from dropest_amd import capi
from dropest_amd.multi import ShardedRun
from dropest_amd.synth import SynthStream
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
R = int(1e8)
run = ShardedRun(SynthStream(n_reads=R, n_cells=5000, n_genes=30000, cb_len=16, umi_len=10, stream_id=2), 0, 1, 0, R,
                 {"min_before": 20, "min_after": 100}, dist)
for _ in range(2):
    run.step()
run.trace = {}
K = 5
t0 = time.perf_counter()
for _ in range(K):
    run.step()
torch.cuda.synchronize()
print("ms/step (traced, extra syncs)", (time.perf_counter() - t0) / K * 1e3)
for k, v in run.trace.items():
    print("  %-22s %7.2f" % (k, v / K))
run.trace = None
run.set_profiling(True)
run.step()
st = run.kernel_stats()
print(sorted([(round(v["ms"], 2), k) for k, v in st.items()], reverse=True)[:25])
import cProfile, pstats, io
run.set_profiling(False)
orig = run._global_columns
def wrapped(everyone, metas, filtered):
    print("global_columns: cells", sum(len(x) for x in everyone), "meta rows", sum(len(m) for m in metas), "filtered", filtered)
    return orig(everyone, metas, filtered)
run._global_columns = wrapped
run.step()
run._global_columns = orig
pr = cProfile.Profile(); pr.enable()
for _ in range(3): run.step()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
