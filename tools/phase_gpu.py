"""Dev tool: GPU time of the step's phases from events on the compute stream (no tracer): forward program, criterion,
backward (criterion's + the model's program), Adam."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "avid-cma_amd")):
    sys.path.insert(0, p)
import torch
import models, criterions
from avid_hip import plan
from avid_hip.parallel import TrainStep
dev = torch.device("cuda:0")
bs, N = 64, 240000
torch.manual_seed(0)
model = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128]).to(dev).train()
crit = criterions.AVID(num_data=N, embedding_dim=128, num_negatives=1024, momentum=0.5, xModal_coeff=1., wModal_coeff=0., device=0)
eng = TrainStep(model, crit)
g = torch.Generator().manual_seed(1)
video = torch.randn(bs, 3, 8, 112, 112, generator=g).to(dev)
audio = torch.randn(bs, 1, 40, 100, generator=g).to(dev)
ids = torch.stack([torch.randperm(N, generator=g)[:bs] for _ in range(64)]).to(dev)
for i in range(8):
    eng.step(video, audio, ids[i])
torch.cuda.synchronize()
n = 30
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(n)]
for i in range(n):
    e = ev[i]
    e[0].record()
    with plan.engine(eng):
        out = plan.run(model, video, audio)
        e[1].record()
        loss, _ = crit(out[0], out[1], ids[8 + i])
        e[2].record()
        loss.backward()
    eng.buckets.finish()
    e[3].record()
    eng.optimizer_step()
    eng._poll_errors()
    e[4].record()
torch.cuda.synchronize()
names = ["forward", "criterion", "backward", "adam"]
tot = [0.0] * 4
for e in ev[5:]:
    for k in range(4):
        tot[k] += e[k].elapsed_time(e[k + 1])
m = n - 5
print("GPU phases (ms): " + "  ".join(f"{names[k]} {tot[k] / m:.3f}" for k in range(4)) + f"   sum {sum(tot) / m:.3f}")
