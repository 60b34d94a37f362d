"""Dev tool: per-kernel time / effective bandwidth of the BatchNorm kernels at the step's activation sizes."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
import torch
from avid_hip import lib, ops
dev = torch.device("cuda:0")
shapes = [(64*8*56*56, 64), (64*8*28*28, 64), (64*4*14*14, 128), (64*2*7*7, 256), (64*1*4*4, 512), (64*20*50, 64)]
reps = 5
for M, C in shapes:
    x = torch.randn(M, C, device=dev).requires_grad_(True)
    g = torch.ones(C, device=dev, requires_grad=True); b = torch.zeros(C, device=dev, requires_grad=True)
    rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev)
    y = ops.batch_norm_cl(x, g, b, rm, rv, True, 0.1, 1e-5, True)
    gy = torch.randn_like(y)
    y.backward(gy); torch.cuda.synchronize()
    lib.timing_enable(True)
    for _ in range(reps):
        x.grad = None
        y = ops.batch_norm_cl(x, g, b, rm, rv, True, 0.1, 1e-5, True)
        y.backward(gy)
    torch.cuda.synchronize()
    rep = lib.timing_report(); lib.timing_enable(False)
    line = f"M={M:8d} C={C:4d} ({M*C*4/1e6:6.1f} MB): "
    for n, v in rep.items():
        us = v["ms"] / reps * 1e3
        line += f"{n.replace('_kernel','')}: {us:6.1f} us {v['bytes']/v['launches']/us/1e3:5.2f} TB/s | "
    print(line)
