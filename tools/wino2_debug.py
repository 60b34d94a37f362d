"""Development: wino2_kernel against torch's convolution, with the positions of the mismatches."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
import torch
import torch.nn.functional as F
from avid_hip import lib, ops
dev = torch.device("cuda:0")
B, T_, H, W = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "4,8,48,48").split(",")]
cin, cout = int(sys.argv[2]) if len(sys.argv) > 2 else 64, int(sys.argv[3]) if len(sys.argv) > 3 else 128
torch.manual_seed(0)
x = torch.randn(B, T_, H, W, cin, device=dev)
w = ops.make_weight(cout, cin, 1, 3, 3).normal_().to(dev)
if os.environ.get("DET"):
    from oracle import detgen
    shape = (B, T_, H, W)
    x = torch.from_numpy(detgen.det_normalish(f"cbp:{shape}:{cin}:x", (B, T_, H, W, cin))).to(dev)
    w = ops.make_weight(cout, cin, 1, 3, 3)
    w.copy_(torch.from_numpy(detgen.det_param(f"cbp:{cout}:{cin}:w.weight", (cout, cin, 1, 3, 3))))
    w = w.to(dev)
ref = F.conv3d(x.permute(0, 4, 1, 2, 3).double(), w.permute(0, 4, 1, 2, 3).double() if w.dim() == 5 and w.shape[-1] == cin else w.double(), padding=(0, 1, 1)).permute(0, 2, 3, 4, 1).float()
ys = []
add = torch.rand_like(ref.contiguous()) if os.environ.get("ADD") else None
if add is not None: ref = ref + add
for stats in (False, True, False, True):
    out = ops.conv_cl(x, w, (1, 1, 1), (0, 1, 1), bn_stats=stats, addend=add)
    y = out[0] if stats else out
    torch.cuda.synchronize()
    ys.append(y.clone())
    err = (y - ref).abs()
    print("stats", stats, "max err", err.max().item(), "ref scale", ref.abs().max().item())
    bad = (err > 1e-3 * ref.abs().max()).nonzero()
    print("  bad count", bad.shape[0], "of", y.numel())
    if bad.shape[0]:
        print("  first bad", bad[:8].tolist())
        print("  frames", bad[:, 0].unique().tolist()[:10], bad[:, 1].unique().tolist()[:10], "rows", bad[:, 2].unique().tolist()[:20], "cols", bad[:, 3].unique().tolist()[:20], "ch", bad[:, 4].unique().tolist()[:70])
for i in range(1, 4):
    d = (ys[i] != ys[0]).nonzero()
    print("run", i, "vs 0: differing", d.shape[0], d[:5].tolist())
