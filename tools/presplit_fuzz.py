"""Dev tool: random shapes through the two kernels of round 5 that split once at the LDS write, each against the form it replaces —
wgrad_group_kernel (wgrad_pre_body: bit-identical) and the video stem's forward (stem_fwd3p_kernel: to rounding, and against
float64).  usage: python tools/presplit_fuzz.py [cases=40] [seed=0]"""
import sys, os, random, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
import torch
import torch.nn.functional as F
from avid_hip import lib, ops
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.manual_seed(1)
bad = 0
# ---- grouped weight gradients
pre = lib.raw("avid_wgrad_pre_configure")
for case in range(cases):
    n = rng.randint(1, 12)
    layers = []
    while len(layers) < n:
        cin, cout = rng.choice([64, 128, 256, 512]), rng.choice([128, 256, 512])
        k, pad = rng.choice([((3, 1, 1), (1, 0, 0)), ((1, 3, 3), (0, 1, 1)), ((1, 1, 1), (0, 0, 0))])
        stride = rng.choice([(1, 1, 1), (2, 1, 1), (1, 2, 2), (2, 2, 2)]) if k != (1, 1, 1) else rng.choice([(1, 1, 1), (2, 2, 2)])
        shape = (rng.randint(1, 9), rng.randint(1, 6), rng.randint(2, 15), rng.randint(2, 15))
        d = ops._desc_cached(shape, cin, cout, k, stride, pad, False)[0]
        if d.groupable:
            layers.append((cin, cout, k, stride, pad, shape, d))
    items = (lib.WgradItem * n)()
    keep, outs = [], []
    for i, (cin, cout, k, stride, pad, (B, T, H, W), d) in enumerate(layers):
        x = torch.randn(B, T, H, W, cin, device=dev)
        To, Ho, Wo = [(a + 2 * p - kk) // s + 1 for a, p, kk, s in zip((T, H, W), pad, k, stride)]
        gy = torch.randn(B, To, Ho, Wo, cout, device=dev)
        dw = ops.make_weight(cout, cin, *k).to(dev)
        items[i].d = d
        items[i].x, items[i].dy, items[i].dw = x.data_ptr(), gy.data_ptr(), dw.data_ptr()
        keep += [x, gy]
        outs.append(dw)
    nb = lib.raw("avid_conv_wgrad_group_workspace_bytes")(n, items)
    ws = torch.empty(max(int(nb), 16), dtype=torch.uint8, device=dev)
    res = {}
    for on in (1, 0):
        pre(on)
        for o in outs:
            o.fill_(float("nan"))
        lib.call("avid_conv_wgrad_group", n, items, ops._p(ws), ws.numel(), ops._stream())
        res[on] = [o.clone() for o in outs]
    pre(-1)
    ok = all(torch.equal(a, b) and bool(torch.isfinite(a).all()) for a, b in zip(res[1], res[0]))
    if not ok:
        bad += 1
        print("wgrad_group MISMATCH", [(l[:6]) for l in layers])
print(f"wgrad_group: {cases} random groups, {bad} mismatches")
# ---- the video stem's forward
spre = lib.raw("avid_stem_fwd_pre_configure")
sbad = 0
for case in range(cases):
    B, T = rng.randint(1, 4), rng.randint(1, 8)
    H, W = 2 * rng.randint(8, 60), 4 * rng.randint(4, 30)
    x = torch.randn(B, 3, T, H, W)
    w = torch.randn(64, 3, 3, 7, 7) * 0.05
    yr = F.conv3d(x.double(), w.double(), stride=(1, 2, 2), padding=(1, 3, 3))
    wd = ops.make_weight(64, 3, 3, 7, 7)
    wd.copy_(w)
    xd, wd = x.to(dev), wd.to(dev)
    ys = {}
    for on in (1, 0):
        spre(on)
        y, part = ops.conv_cl(xd, wd, (1, 2, 2), (1, 3, 3), channel_first=True, bn_stats=True)
        ys[on] = y.permute(0, 4, 1, 2, 3).cpu()
    spre(-1)
    e = [float((ys[on].double() - yr).pow(2).mean().sqrt() / yr.pow(2).mean().sqrt()) for on in (1, 0)]
    dd = float((ys[1] - ys[0]).abs().max() / yr.abs().max())
    if e[0] > 6e-7 or dd > 2e-6:
        sbad += 1
        print("stem MISMATCH", (B, T, H, W), e, dd)
print(f"stem forward: {cases} random shapes, {sbad} outside the bars (6e-7 rms against float64, 2e-6 between the forms)")
sys.exit(1 if bad or sbad else 0)
