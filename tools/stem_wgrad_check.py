import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/avid-cma_amd")
import torch
from avid_hip import ops, lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, T, H, ref64) in ((2, 4, 64, True), (4, 8, 112, True), (64, 8, 112, False), (8, 8, 224, False)):
    x = torch.randn(B, 3, T, H, H, device=dev)
    w = ops.make_weight(64, 3, 3, 7, 7).to(dev)
    w.copy_(torch.randn(64, 3, 3, 7, 7, device=dev) * 0.05)
    w.requires_grad_(True)
    y = ops.conv_cl(x, w, (1, 2, 2), (1, 3, 3), channel_first=True)
    gy = torch.randn_like(y)
    y.backward(gy)
    torch.cuda.synchronize()
    got = w.grad.detach().clone()
    msg = ""
    if ref64:
        xd = x.double().cpu()
        wd = w.detach().double().cpu().contiguous().requires_grad_(True)
        yd = torch.nn.functional.conv3d(xd, wd, stride=(1, 2, 2), padding=(1, 3, 3))
        yd.backward(gy.double().cpu().permute(0, 4, 1, 2, 3))
        ref = wd.grad
        err = (got.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
        rms = ((got.double().cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
        msg = f"max err {err:.3e} rms {rms:.3e}"
    w.grad = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        y = ops.conv_cl(x, w, (1, 2, 2), (1, 3, 3), channel_first=True)
        y.backward(gy)
        w.grad = None
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 10 * 1e6
    print(f"B{B} T{T} {H}x{H}: {msg}  fwd+wgrad {us:.1f} us  |dw| {got.abs().max().item():.3e} split={os.environ.get('AVID_STEM_BF16X3','1')}")
