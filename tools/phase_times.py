"""Dev tool: GPU time of the step's phases (events on the main stream), optionally on a 1-rank RCCL group."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
import torch, models, criterions
from avid_hip.parallel import TrainStep
dev = torch.device("cuda:0")
if os.environ.get("AVID_FORCE_DIST") == "1" or os.environ.get("AVID_PG_ONLY"):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29513")
    dist.init_process_group(os.environ.get("AVID_PG_ONLY") or "nccl", rank=0, world_size=1, **({"device_id": dev} if os.environ.get("AVID_PG_ONLY", "nccl") == "nccl" and not os.environ.get("AVID_PG_LAZY") else {}))
m = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128]).to(dev).train()
c = criterions.AVID(num_data=240000, embedding_dim=128, num_negatives=1024, momentum=0.5, device=0)
e = TrainStep(m, c)
v = torch.randn(64, 3, 8, 112, 112, device=dev); a = torch.randn(64, 1, 40, 100, device=dev)
y = torch.randperm(240000)[:64].to(dev)
for _ in range(5): e.step(v, a, y)
torch.cuda.synchronize()
names = ["zero", "fwd", "crit", "bwd", "finish", "adam"]
acc = {n: 0.0 for n in names}
N = 20
for _ in range(N):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
    ev[0].record(); e.flat.zero_grad()
    ev[1].record(); ve, ae = e.model(v, a)
    ev[2].record(); loss, _ = e.criterion(ve, ae, y)
    ev[3].record()
    e.twt.refresh()
    with e.twt.armed(), e.slots.armed():
        loss.backward()
    ev[4].record(); e.buckets.finish()
    ev[5].record(); e.optimizer_step()
    ev[6].record()
    torch.cuda.synchronize()
    for i, n in enumerate(names): acc[n] += ev[i].elapsed_time(ev[i + 1])
print("  ".join(f"{n} {acc[n]/N:.3f}" for n in names), " total %.3f ms" % (sum(acc.values()) / N))
