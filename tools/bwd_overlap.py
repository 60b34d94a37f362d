"""Dev tool: backward (dgrad + wgrad) of one conv layer, serial vs wgrad on a side stream (AVID_OVERLAP_WGRAD)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
import torch
from avid_hip import ops
dev = torch.device("cuda:0")
L = [("c3.spt", 128, 128, (1,3,3), (1,1,1), (0,1,1), (4,14,14)), ("c4.spt", 256, 256, (1,3,3), (1,1,1), (0,1,1), (2,7,7)),
     ("c4.tmp", 256, 256, (3,1,1), (1,1,1), (1,0,0), (2,7,7)), ("c5.spt", 512, 512, (1,3,3), (1,1,1), (0,1,1), (1,4,4)),
     ("c5.tmp", 512, 512, (3,1,1), (1,1,1), (1,0,0), (1,4,4))]
for name, cin, cout, k, st, pd, (T, H, W) in L:
    x = torch.randn(64, T, H, W, cin, device=dev).requires_grad_(True)
    w = ops.make_weight(cout, cin, *k).normal_().to(dev).requires_grad_(True)
    y = ops.conv_cl(x, w, st, pd); g = torch.randn_like(y)
    for _ in range(3): y.backward(g, retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n):
        x.grad = None; w.grad = None
        y.backward(g, retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1)/n*1e3:7.1f} us per backward (overlap={os.environ.get('AVID_OVERLAP_WGRAD','0')})")
