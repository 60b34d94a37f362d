"""Per-layer conv micro-benchmark (dev tool): TF/s of fwd / dgrad / wgrad for every distinct conv shape
of the R(2+1)D-18 + Conv2D step at a given batch, timed with the library's HIP-event timers."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
import torch
from avid_hip import lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
only = sys.argv[2] if len(sys.argv) > 2 else ""
reps = 5
dev = torch.device("cuda:0")
# name, Cin, Cout, k, stride, pad, (T,H,W) in
L = [
 ("c2.spt", 64, 64, (1,3,3), (1,1,1), (0,1,1), (8,28,28)),
 ("c2.tmp", 64, 64, (3,1,1), (1,1,1), (1,0,0), (8,28,28)),
 ("c3.spt_s2", 64, 128, (1,3,3), (1,2,2), (0,1,1), (8,28,28)),
 ("c3.tmp_s2", 128, 128, (3,1,1), (2,1,1), (1,0,0), (8,14,14)),
 ("c3.res", 64, 128, (1,1,1), (2,2,2), (0,0,0), (8,28,28)),
 ("c3.spt", 128, 128, (1,3,3), (1,1,1), (0,1,1), (4,14,14)),
 ("c3.tmp", 128, 128, (3,1,1), (1,1,1), (1,0,0), (4,14,14)),
 ("c4.spt_s2", 128, 256, (1,3,3), (1,2,2), (0,1,1), (4,14,14)),
 ("c4.spt", 256, 256, (1,3,3), (1,1,1), (0,1,1), (2,7,7)),
 ("c4.tmp", 256, 256, (3,1,1), (1,1,1), (1,0,0), (2,7,7)),
 ("c5.spt_s2", 256, 512, (1,3,3), (1,2,2), (0,1,1), (2,7,7)),
 ("c5.spt", 512, 512, (1,3,3), (1,1,1), (0,1,1), (1,4,4)),
 ("c5.tmp", 512, 512, (3,1,1), (1,1,1), (1,0,0), (1,4,4)),
 ("c4.tmp_s2", 256, 256, (3,1,1), (2,1,1), (1,0,0), (4,7,7)),
 ("c4.res", 128, 256, (1,1,1), (2,2,2), (0,0,0), (4,14,14)),
 ("c5.tmp_s2", 512, 512, (3,1,1), (2,1,1), (1,0,0), (2,4,4)),
 ("c5.res", 256, 512, (1,1,1), (2,2,2), (0,0,0), (2,7,7)),
 ("a.b1_s2", 64, 64, (1,3,3), (1,2,2), (0,1,1), (1,20,50)),
 ("a.b1", 64, 64, (1,3,3), (1,1,1), (0,1,1), (1,10,25)),
 ("a.b2_s2", 64, 128, (1,3,3), (1,2,2), (0,1,1), (1,10,25)),
 ("a.b2", 128, 128, (1,3,3), (1,1,1), (0,1,1), (1,5,13)),
 ("a.b3_s2", 128, 256, (1,3,3), (1,2,2), (0,1,1), (1,5,13)),
 ("a.b3", 256, 256, (1,3,3), (1,1,1), (0,1,1), (1,3,7)),
 ("a.b4a", 256, 512, (1,3,3), (1,1,1), (0,1,1), (1,3,7)),
 ("a.b4", 512, 512, (1,3,3), (1,1,1), (0,1,1), (1,3,7)),
 ("g576", 576, 64, (1,1,1), (1,1,1), (0,0,0), (8,28,28)),
 ("g1152", 1152, 128, (1,1,1), (1,1,1), (0,0,0), (4,14,14)),
 ("r2.spt", 64, 64, (1,3,3), (1,1,1), (0,1,1), (8,56,56)),          # real-dataset shapes (224 x 224 input): run with B = 32
 ("r3.spt", 128, 128, (1,3,3), (1,1,1), (0,1,1), (4,28,28)),
 ("big128", 128, 128, (1,3,3), (1,1,1), (0,1,1), (8,28,28)),      # not a layer of the model: wide vs narrow tile at equal M
]
print(f"{'layer':10s} {'M':>8s} {'K':>5s} {'N':>4s} | {'fwd us':>8s} {'TF':>6s} | {'dgrad us':>8s} {'TF':>6s} | {'wgrad us':>8s} {'TF':>6s}  kernels")
for name, cin, cout, k, st, pd, (T, H, W) in L:
    if only and only not in name: continue
    x = torch.randn(B, T, H, W, cin, device=dev)
    # CB_DATA: operand values change the matrix pipe's power and with it the clock the chip holds
    if os.environ.get("CB_DATA") == "zero": x.zero_()
    elif os.environ.get("CB_DATA") == "relu": x.relu_()
    x.requires_grad_(True)
    w = ops.make_weight(cout, cin, *k).normal_().to(dev).requires_grad_(True)
    y = ops.conv_cl(x, w, st, pd)
    g = torch.randn_like(y)
    y.backward(g)
    torch.cuda.synchronize()
    res = {}
    for which in ("fwd", "bwd"):
        lib.timing_enable(True)
        for _ in range(reps):
            if which == "fwd":
                y = ops.conv_cl(x, w, st, pd)
            else:
                x.grad = None; w.grad = None
                y.backward(g, retain_graph=True)
        torch.cuda.synchronize()
        res[which] = lib.timing_report()
        lib.timing_enable(False)
    if os.environ.get('AVID_DBG') == '8':
        v = y.flatten()[:2].tolist(); print(f'   clock: {v[0]:.0f} shader ticks / {v[1]:.0f} x10ns = {v[0]/(v[1]/100):.0f} MHz')
    if os.environ.get('CB_VERBOSE'):
        for which in ('fwd','bwd'):
            for n, v in res[which].items(): print(f'      {which} {n:34s} {v["ms"]/reps*1e3:8.1f} us  x{v["launches"]/reps:.0f}')
    M = y.numel() // cout
    K = cin * k[0] * k[1] * k[2]
    fl = 2.0 * M * cout * K
    def pick(rep, pref, mode=None):
        t = 0.0; names = []
        for n, v in rep.items():
            if n.startswith(pref) and (mode is None or n.endswith(f",{mode}>") or n.endswith(f",{mode}>s2")):
                t += v["ms"]; names.append(n)
        return t / reps * 1e3, names
    def pick_w(rep, mode):     # the Winograd path: kernel + its weight transform
        t = 0.0; names = []
        for n, v in rep.items():
            if n.startswith(("wino_kernel<", "wino2_kernel<", "wino2p_kernel<", "winot_kernel<")) or n == "wino_weight_kernel":
                t += v["ms"]; names.append(n)
        return t / reps * 1e3, [n for n in names if n.startswith(("wino_kernel", "wino2_kernel", "wino2p_kernel", "winot_kernel"))]
    f_us, fn = pick(res["fwd"], "igemm_", 0)
    tf_us, tfn = pick(res["fwd"], "tconv64_kernel<0>")
    f_us += tf_us; fn += tfn
    wf_us, wfn = pick_w(res["fwd"], 0)
    f_us += wf_us; fn += wfn
    fr_us, _ = pick(res["fwd"], "splitk")
    d_us, dn = pick(res["bwd"], "igemm_", 1)
    td_us, tdn = pick(res["bwd"], "tconv64_kernel<1>")
    d_us += td_us; dn += tdn
    wd_us, wdn = pick_w(res["bwd"], 1)
    d_us += wd_us; dn += wdn
    dr_us, _ = pick(res["bwd"], "splitk"); dt_us, _ = pick(res["bwd"], "weight_tr")
    w_us, wn = pick(res["bwd"], "wgrad_tab")
    tw_us, twn = pick(res["bwd"], "twgrad64")
    w_us += tw_us; wn += twn
    ww_us, wwn = pick(res["bwd"], "wino_wgrad")
    w_us += ww_us; wn += wwn
    wr_us, _ = pick(res["bwd"], "wgrad_reduce")
    print(f"{name:10s} {M:8d} {K:5d} {cout:4d} | {f_us+fr_us:8.1f} {fl/(f_us+fr_us)/1e6:6.1f} | {d_us+dr_us+dt_us:8.1f} {fl/(d_us+dr_us+dt_us)/1e6:6.1f} | "
          f"{w_us+wr_us:8.1f} {fl/(w_us+wr_us)/1e6:6.1f}  {fn+dn+wn} (+red {fr_us:.0f}/{dr_us+dt_us:.0f}/{wr_us:.0f})")
