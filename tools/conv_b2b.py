"""Dev tool: back-to-back timing of one conv forward (torch events around N launches, no per-kernel timers)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
import torch
from avid_hip import ops
dev = torch.device("cuda:0")
cases = {
 "g576": (576, 64, (1,1,1), (1,1,1), (0,0,0), (8,28,28)),
 "c2.spt": (64, 64, (1,3,3), (1,1,1), (0,1,1), (8,28,28)),
 "c3.spt": (128, 128, (1,3,3), (1,1,1), (0,1,1), (4,14,14)),
 "g1152": (1152, 128, (1,1,1), (1,1,1), (0,0,0), (4,14,14)),
 "c2.spt@224": (64, 64, (1,3,3), (1,1,1), (0,1,1), (8,56,56)),
 "c2.tmp@224": (64, 64, (3,1,1), (1,1,1), (1,0,0), (8,56,56)),
}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, (cin, cout, k, st, pd, (T, H, W)) in cases.items():
    x = torch.randn(B, T, H, W, cin, device=dev)
    w = ops.make_weight(cout, cin, *k).normal_().to(dev)
    with torch.no_grad():
        for _ in range(3): y = ops.conv_cl(x, w, st, pd)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n): y = ops.conv_cl(x, w, st, pd)
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    M = y.numel() // cout; K = cin * k[0] * k[1] * k[2]
    print(f"{name:8s} M={M} K={K} N={cout}: {us:8.1f} us  {2.0*M*cout*K/us/1e6:6.1f} TF (launches back to back, incl. reduce)")
