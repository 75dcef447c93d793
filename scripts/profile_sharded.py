"""Developer tool: cProfile of the sharded step (world = 1, RCCL) on one GPU -- shows host-side orchestration costs."""
import os, sys, cProfile, pstats, io, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
os.environ["RANK"]="0"; os.environ["WORLD_SIZE"]="1"
import torch, torch.distributed as dist
from dropest_amd import capi
from dropest_amd.multi import ShardedRun
from dropest_amd.synth import SynthStream
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda",0))
cfgname = sys.argv[1] if len(sys.argv)>1 else "c2"
cfg = {"min_before":20,"min_after":100}
R = int(1e8)
kw = dict(n_reads=R, n_cells=5000, n_genes=30000, cb_len=16, umi_len=10, stream_id=2)
if cfgname=="c4":
    R=int(1.25e8); kw=dict(n_reads=R,n_cells=5000,n_genes=30000,cb_len=16,umi_len=8,stream_id=4,whitelist="indrop_v3")
    cfg["merge"]={"barcodes_kind":capi.BARCODES_CONST,"barcodes_file":os.path.join("dropest_amd","data","barcodes","indrop_v3")}
run = ShardedRun(SynthStream(**kw), 0, 1, 0, R, cfg, dist)
for _ in range(2): run.step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0=time.perf_counter()
for _ in range(5): run.step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter()-t0)/5*1e3)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:6000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_callers("_cuda_synchronize|synchronize"); print(s.getvalue()[:3000])
