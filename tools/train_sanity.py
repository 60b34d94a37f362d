"""Development: 200 optimisation steps on a fixed set of 512 synthetic clips — the loss must fall and stay finite (a whole-path
sanity run of the kernels inside a real optimisation, not a parity test)."""
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/avid-cma_amd")
import torch, models, criterions
from avid_hip.parallel import TrainStep
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128]).to(dev).train()
N = 512
c = criterions.AVID(num_data=N, embedding_dim=128, num_negatives=256, momentum=0.5, device=0)
e = TrainStep(m, c, lr=1e-3, weight_decay=1e-5)
g = torch.Generator().manual_seed(1)
# a tiny fixed dataset of 512 clips in 8 batches of 64: the loss must fall as the banks and the towers fit it
vids = [torch.randn(64, 3, 8, 112, 112, generator=g).to(dev) for _ in range(8)]
auds = [torch.randn(64, 1, 40, 100, generator=g).to(dev) for _ in range(8)]
ids = [torch.arange(64 * i, 64 * (i + 1), device=dev) for i in range(8)]
hist = []
for ep in range(25):
    tot = 0.0
    for i in range(8):
        tot += float(e.step(vids[i], auds[i], ids[i]))
    hist.append(tot / 8)
    if ep % 4 == 0 or ep == 24: print(f"epoch {ep:2d} loss {hist[-1]:.4f}", flush=True)
assert all(x == x for x in hist), "NaN"
assert hist[-1] < hist[1] - 0.3, hist
print("ok: loss fell from", round(hist[1], 3), "to", round(hist[-1], 3))
