"""Dev tool: GPU time (host-paced) of the AVID_CMA criterion alone at 240k rows, forward and backward."""
import os, sys
REPO = "/root/repo"
for p in (REPO, os.path.join(REPO, "avid-cma_amd")): sys.path.insert(0, p)
import torch, criterions
from avid_hip import lib
dev = torch.device("cuda:0")
bs, N = 64, 240000
torch.manual_seed(0)
crit = criterions.AVID_CMA(num_data=N, embedding_dim=128, num_negatives=1024, num_negatives_within=None, momentum=0.5,
                           xModalInstCoeff=1., wModalInstCoeff=0., xModalPosCoeff=0., wModalPosCoeff=1.,
                           sampling_args={"type": "consensus", "pos_k": 32}, device=0)
g = torch.Generator().manual_seed(1)
ids = torch.stack([torch.randperm(N, generator=g)[:bs] for _ in range(40)]).to(dev)
v = torch.randn(bs, 128, device=dev, requires_grad=True); a = torch.randn(bs, 128, device=dev, requires_grad=True)
for i in range(5):
    loss, _ = crit(v, a, ids[i]); loss.backward()
torch.cuda.synchronize()
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(20)]
for i in range(20):
    ev[i][0].record(); loss, _ = crit(v, a, ids[5 + i]); ev[i][1].record(); loss.backward(); ev[i][2].record()
torch.cuda.synchronize()
f = sorted(e[0].elapsed_time(e[1]) for e in ev)[10]; b = sorted(e[1].elapsed_time(e[2]) for e in ev)[10]
print(f"AVID_CMA criterion GPU+host-paced: forward {f:.3f} ms, backward {b:.3f} ms")
