"""Dev tool: the training step at the shipped configs' REAL input shapes: 3x8x224x224 video,
1x200x257 audio (configs/main/avid/kinetics/Cross-N1024.yaml:19-25), per-kernel HIP-event table.
usage: python tools/real_shape_bench.py [batch=32] [steps=8]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
import torch, models, criterions
from avid_hip import lib
from avid_hip.parallel import TrainStep
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
m = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128]).to(dev).train()
c = criterions.AVID(num_data=240000, embedding_dim=128, num_negatives=1024, momentum=0.5, device=0)
e = TrainStep(m, c)
v = torch.randn(B, 3, 8, 224, 224, device=dev); a = torch.randn(B, 1, 200, 257, device=dev)
y = torch.randperm(240000)[:B].to(dev)
for _ in range(3): e.step(v, a, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): e.step(v, a, y)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
# 4x the video FLOPs of the 112^2 benchmark shape, ~6.4x the audio's
print(f"batch {B}: {dt*1e3:.2f} ms/step  {B/dt:.1f} clips/s")
m.overlap_towers = False
lib.timing_enable(True)
for _ in range(2): e.step(v, a, y)
torch.cuda.synchronize()
k = lib.timing_report(); lib.timing_enable(False)
tot = sum(x["ms"] for x in k.values())
for n, x in sorted(k.items(), key=lambda kv: -kv[1]["ms"])[:16]:
    tf = x["flops"] / (x["ms"] * 1e-3) / 1e12 if x["flops"] else 0
    print(f"  {n:34s} {x['launches']/2:6.1f}/step {x['ms']/2:8.3f} ms {100*x['ms']/tot:5.1f}%  {tf:7.1f} TF")
