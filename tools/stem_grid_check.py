"""Dev tool: the video stem's forward at several CU budgets (grids) in both forms: outputs bit-identical, partial sums agree."""
import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/avid-cma_amd")
from avid_hip import lib, ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(4, 3, 8, 112, 112, device=dev)
w = ops.make_weight(64, 3, 3, 7, 7).normal_().to(dev)
res = {}
for pre in (1, 0):
    lib.raw("avid_stem_fwd_pre_configure")(pre)
    for budget in (0, 248, 200, 0):
        lib.raw("avid_set_cu_budget")(budget)
        outs = []
        for rep in range(3):
            y, part = ops.conv_cl(x, w, (1, 2, 2), (1, 3, 3), channel_first=True, bn_stats=True)
            rows = int(lib.raw("avid_cu_budget")()) or part.shape[0]          # the launch writes one row per workgroup
            outs.append((y.clone(), part[:rows].double().sum(0).clone()))
        torch.cuda.synchronize()
        same = all(torch.equal(outs[0][0], o[0]) for o in outs)
        key = (pre, budget)
        if (pre, 0) in res:
            d = (outs[0][0] - res[(pre, 0)][0]).abs().max().item()
            ds = ((outs[0][1] - res[(pre, 0)][1]).abs().max() / res[(pre, 0)][1].abs().max()).item()
        else:
            d = ds = 0.0
            res[key] = outs[0]
        print(f"pre {pre} budget {budget}: repeat-identical {same}, max|y - y(budget 0)| {d:.3e}, stats rel diff {ds:.3e}, rows {part.shape[0]}, nan {bool(torch.isnan(outs[0][0]).any())}")
