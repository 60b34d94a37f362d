"""Development: random shapes through the Winograd forward / input-gradient / weight-gradient kernels (both
forward kernels, statistics and addend epilogues) against float64 F.conv3d.  usage: python tools/wino_fuzz.py [n] [seed]"""
import os, sys, random
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
import torch
import torch.nn.functional as F
from avid_hip import ops
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ops.wino_configure(1, 1, 256)
worst = 0.0
for it in range(n):
    cin, cout = rng.choice([64, 128, 192, 256]), rng.choice([64, 128, 192, 256])
    B, T_ = rng.randint(1, 5), rng.randint(1, 4)
    H, W = rng.randint(2, 40), rng.randint(2, 40)
    v2 = rng.random() < 0.6
    ops.wino2_configure(0 if v2 else 100000)
    g = torch.Generator().manual_seed(it)
    x = torch.randn(B, T_, H, W, cin, generator=g).to(dev).requires_grad_(True)
    w = ops.make_weight(cout, cin, 1, 3, 3)
    w.copy_(torch.randn(cout, cin, 1, 3, 3, generator=g))
    w = w.to(dev).requires_grad_(True)
    add = torch.randn(B, T_, H, W, cout, generator=g).to(dev) if rng.random() < 0.5 else None
    stats = rng.random() < 0.5
    out = ops.conv_cl(x, w, (1, 1, 1), (0, 1, 1), addend=add, bn_stats=stats)
    y, part = out if stats else (out, None)
    gy = torch.randn(y.shape, generator=g).to(dev)
    y.backward(gy)
    xr = x.detach().double().permute(0, 4, 1, 2, 3).requires_grad_(True)
    wr = w.detach().double().requires_grad_(True)
    yr = F.conv3d(xr, wr, padding=(0, 1, 1)).permute(0, 2, 3, 4, 1)
    if add is not None: yr = yr + add.double()
    (yr * gy.double()).sum().backward()
    def rel(a, b): return float((a.double() - b).abs().max() / (b.abs().max() + 1e-30))
    e = [rel(y.detach(), yr.detach()), rel(x.grad, xr.grad.permute(0, 2, 3, 4, 1)), rel(w.grad, wr.grad)]
    if stats and part is not None and part.numel():
        yd = y.detach().double().reshape(-1, cout)
        e.append(rel(part[:, 0].double().sum(0), yd.sum(0)) * float(yd.sum(0).abs().max() / (yd.abs().sum(0).max() + 1e-30)))
        e.append(rel(part[:, 1].double().sum(0), (yd * yd).sum(0)))
    worst = max(worst, max(e))
    flag = "" if max(e) < 5e-5 else "   <-- FAIL"
    print(f"{it:3d} {'wino2' if v2 else 'wino '} B{B} T{T_} {H:2d}x{W:2d} {cin:3d}->{cout:3d} add={add is not None!s:5s} stats={stats!s:5s} " + " ".join(f"{v:.1e}" for v in e) + flag)
print("worst", worst)
ops.wino_configure(-1, -1, -1); ops.wino2_configure(-1)
