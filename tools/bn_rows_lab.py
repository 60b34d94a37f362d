"""Dev tool: how the fused finalize + apply launches of the small layers (bn_fin_apply_kernel / bn_bwd_fin_apply_kernel) depend on
the number of partial rows they fold (the producing convolution's workgroups + reduce blocks)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
import torch
from avid_hip import lib, ops
dev = torch.device("cuda:0")
reps = 20
for M, C in ((1024, 512), (1344, 512), (6272, 256), (50176, 128), (16000, 64)):
    x = torch.randn(M, C, device=dev)
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev)
    line = f"M={M:6d} C={C:4d}: "
    for nparts in (0, 8, 64, 256, 512, 1024):
        part = None
        if nparts:
            part = torch.zeros(nparts, 2, C, device=dev)
            part[0, 0] = x.sum(0); part[0, 1] = (x * x).sum(0)
        for _ in range(3): ops.batch_norm_cl(x, g, b, rm, rv, True, 0.1, 1e-5, True, partials=part)
        torch.cuda.synchronize()
        lib.timing_enable(True)
        for _ in range(reps): ops.batch_norm_cl(x, g, b, rm, rv, True, 0.1, 1e-5, True, partials=part)
        torch.cuda.synchronize()
        rep = lib.timing_report(); lib.timing_enable(False)
        us = sum(v["ms"] for v in rep.values()) / reps * 1e3
        line += f"rows {nparts:4d}: {us:5.1f} us | "
    print(line)
