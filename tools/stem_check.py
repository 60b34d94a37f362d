import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/avid-cma_amd")
import torch
from avid_hip import ops, lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, T, H) in ((2, 8, 64), (4, 8, 112), (64, 8, 112), (8, 8, 224)):
    x = torch.randn(B, 3, T, H, H, device=dev)
    w = ops.make_weight(64, 3, 3, 7, 7).to(dev)
    w.copy_(torch.randn(64, 3, 3, 7, 7, device=dev) * 0.05)
    y = ops.conv_cl(x, w, (1, 2, 2), (1, 3, 3), channel_first=True, bn_stats=True)
    yy, st = y
    ref = torch.nn.functional.conv3d(x.double(), w.double(), stride=(1, 2, 2), padding=(1, 3, 3)).permute(0, 2, 3, 4, 1)
    err = (yy.double() - ref).abs().max().item() / ref.abs().max().item()
    rms = ((yy.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    s_ref = ref.reshape(-1, 64).sum(0)
    s_got = st[:, 0].double().sum(0) if st.numel() else None
    serr = ((s_got - s_ref).abs().max() / s_ref.abs().max()).item() if s_got is not None else -1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ops.conv_cl(x, w, (1, 2, 2), (1, 3, 3), channel_first=True, bn_stats=True)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 10 * 1e6
    print(f"B{B} T{T} {H}x{H}: max err {err:.3e} rms {rms:.3e} stats err {serr:.3e}  {us:.1f} us  split={os.environ.get('AVID_STEM_BF16X3','1')}")
