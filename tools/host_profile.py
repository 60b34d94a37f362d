"""Dev tool: where does the host time of one eager step go? (cProfile over 10 steps)"""
import cProfile, pstats, os, sys, io
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
import torch, models, criterions
from avid_hip.parallel import TrainStep
dev = torch.device("cuda:0")
if os.environ.get("AVID_FORCE_DIST") == "1":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
m = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128]).to(dev).train()
c = criterions.AVID(num_data=240000, embedding_dim=128, num_negatives=1024, momentum=0.5, device=0)
e = TrainStep(m, c)
v = torch.randn(64, 3, 8, 112, 112, device=dev); a = torch.randn(64, 1, 40, 100, device=dev)
y = torch.randperm(240000)[:64].to(dev)
for _ in range(3): e.step(v, a, y)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(10): e.step(v, a, y)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host issue {1e3*(t1-t0)/10:.2f} ms/step, +sync {1e3*(t2-t0)/10:.2f}")
pr = cProfile.Profile(); pr.enable()
for _ in range(10): e.step(v, a, y)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
