import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/avid-cma_amd")
import torch
from avid_hip import ops
dev = torch.device("cuda:0")
x = torch.randn(64, 3, 8, 112, 112, device=dev)
w = ops.make_weight(64, 3, 3, 7, 7).to(dev)
w.copy_(torch.randn(64, 3, 3, 7, 7, device=dev) * 0.05)
for _ in range(4):
    ops.conv_cl(x, w, (1, 2, 2), (1, 3, 3), channel_first=True, bn_stats=True)
torch.cuda.synchronize()
