"""Ad-hoc GPU diagnostics (dev tool, not part of the product path)."""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
from oracle import avid_oracle as O, detgen
import models
from avid_hip import ops

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
dev = torch.device("cuda:0")

def grads_table():
    spec = O.av_wrapper_spec(18)
    P = O.det_state(spec, "w")
    m = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128])
    m.load_state_dict({k: v.clone() for k, v in P.items()})
    m = m.to(dev).train()
    video = T(detgen.det_normalish("in:video", (2, 3, 8, 112, 112)))
    audio = T(detgen.det_normalish("in:audio", (2, 1, 40, 100)))
    gv, ga = T(detgen.det_uniform("in:gv", (2, 128))), T(detgen.det_uniform("in:ga", (2, 128)))
    pn = [n for n in P if not ("running" in n or "num_batches" in n)]
    for n in pn: P[n].requires_grad_(True)
    ve, ae = O.av_forward(video, audio, P, 18, True)
    ((ve * gv).sum() + (ae * ga).sum()).backward()
    e1, e2 = m(video.to(dev), audio.to(dev))
    ((e1 * gv.to(dev)).sum() + (e2 * ga.to(dev)).sum()).backward()
    print("emb err", float((e1.cpu()-ve).abs().max()/ve.abs().max()), float((e2.cpu()-ae).abs().max()/ae.abs().max()))
    for n, p in m.named_parameters():
        a, r = p.grad.contiguous().cpu().double(), P[n].grad.double()
        print(f"{n:50s} {float((a-r).abs().max()/(r.abs().max()+1e-30)):.3e}  norm {float(r.norm()):.3e}")

def bn_case():
    M, C = 12544, 64
    x = T(detgen.det_normalish(f"bn:{M}:{C}:x", (M, C))) * 1.7 + 0.3
    g = T(detgen.det_param(f"bn:{M}:{C}:bn.weight", (C,)))
    b = T(detgen.det_param(f"bn:{M}:{C}:bn.bias", (C,)))
    gy = T(detgen.det_uniform(f"bn:{M}:{C}:gy", (M, C)))
    xr, gr, br = (t.double().requires_grad_(True) for t in (x, g, b))
    yr0 = F.batch_norm(xr.t().unsqueeze(0), None, None, gr, br, True, 0.1, 1e-5).squeeze(0).t()
    yr = F.relu(yr0)
    (yr * gy.double()).sum().backward()
    xd, gd, bd = (t.to(dev).requires_grad_(True) for t in (x, g, b))
    y = ops.batch_norm_cl(xd.view(1, 1, 1, M, C), gd, bd, torch.zeros(C, device=dev), torch.ones(C, device=dev), True, 0.1, 1e-5, True)
    y.backward(gy.to(dev).view(1, 1, 1, M, C))
    d = (xd.grad.cpu().double() - xr.grad).abs()
    print("bn dx: max diff", float(d.max()), "count > 1e-4:", int((d > 1e-4).sum()), "of", d.numel())
    bad = (d > 1e-4).nonzero()
    for i in bad[:10]:
        r, c = int(i[0]), int(i[1])
        print(r, c, "yref_pre", float(yr0[r, c]), "ygpu", float(y.view(M, C)[r, c]), "dx", float(xd.grad[r, c]), float(xr.grad[r, c]))
    per_ch = d.max(0).values
    print("channels with err>1e-4:", (per_ch > 1e-4).nonzero().flatten().tolist())

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("bn", "all"): bn_case()
    if which in ("grads", "all"): grads_table()


def block_debug(cin=512, cout=512, shape=(2, 512, 1, 4, 4)):
    """identity R(2+1)D block at T=1: compare every intermediate grad against torch CPU."""
    from models.network_blocks import BasicR2P1DBlock
    blk = BasicR2P1DBlock(cin, cout)
    sd = {k: T(detgen.det_param(f"dbg:{k}", tuple(v.shape)).copy()).to(v.dtype) for k, v in blk.state_dict().items()}
    blk.load_state_dict(sd)
    blk = blk.to(dev).train()
    x = T(detgen.det_normalish("dbg:x", shape))
    P = {f"b.{k}": v.clone() for k, v in sd.items()}
    for k in P:
        if k.endswith("weight") or k.endswith("bias"): P[k].requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    inter_r = {}
    # reference, step by step
    h1 = F.conv3d(xr, P["b.spt_conv1.weight"], padding=(0, 1, 1)); h1.retain_grad(); inter_r["c1"] = h1
    a1 = F.relu(F.batch_norm(h1, None, None, P["b.spt_bn1.weight"], P["b.spt_bn1.bias"], True)); a1.retain_grad(); inter_r["a1"] = a1
    h2 = F.conv3d(a1, P["b.tmp_conv1.weight"], padding=(1, 0, 0)); h2.retain_grad(); inter_r["c2"] = h2
    a2 = F.relu(F.batch_norm(h2, None, None, P["b.tmp_bn1.weight"], P["b.tmp_bn1.bias"], True)); a2.retain_grad(); inter_r["a2"] = a2
    h3 = F.conv3d(a2, P["b.spt_conv2.weight"], padding=(0, 1, 1)); h3.retain_grad(); inter_r["c3"] = h3
    a3 = F.relu(F.batch_norm(h3, None, None, P["b.spt_bn2.weight"], P["b.spt_bn2.bias"], True)); a3.retain_grad(); inter_r["a3"] = a3
    h4 = F.conv3d(a3, P["b.tmp_conv2.weight"], padding=(1, 0, 0)) + xr; h4.retain_grad(); inter_r["s"] = h4
    out = F.relu(F.batch_norm(h4, None, None, P["b.out_bn.weight"], P["b.out_bn.bias"], True))
    gy = T(detgen.det_uniform("dbg:g", tuple(out.shape)))
    (out * gy).sum().backward()
    # gpu
    cl = lambda t: t.permute(0, 2, 3, 4, 1).contiguous()
    nc = lambda t: t.permute(0, 4, 1, 2, 3)
    xd = cl(x).to(dev).requires_grad_(True)
    inter = {}
    c1 = blk.spt_conv1(xd); c1.retain_grad(); inter["c1"] = c1
    a1 = blk.spt_bn1(c1, relu=True); a1.retain_grad(); inter["a1"] = a1
    c2 = blk.tmp_conv1(a1); c2.retain_grad(); inter["c2"] = c2
    a2 = blk.tmp_bn1(c2, relu=True); a2.retain_grad(); inter["a2"] = a2
    c3 = blk.spt_conv2(a2); c3.retain_grad(); inter["c3"] = c3
    a3 = blk.spt_bn2(c3, relu=True); a3.retain_grad(); inter["a3"] = a3
    s = blk.tmp_conv2(a3, addend=xd); s.retain_grad(); inter["s"] = s
    o = blk.out_bn(s, relu=True)
    o.backward(cl(gy).to(dev))
    e = lambda a, r: float((a.double().cpu() - r.double()).abs().max() / (r.double().abs().max() + 1e-30))
    print("out", e(nc(o.detach()), out.detach()))
    for k in ["s", "a3", "c3", "a2", "c2", "a1", "c1"]:
        print(k, "val", e(nc(inter[k].detach()), inter_r[k].detach()), "grad", e(nc(inter[k].grad), inter_r[k].grad))
    print("x grad", e(nc(xd.grad), xr.grad))
    for n, p in blk.named_parameters():
        print(n, e(p.grad.contiguous(), P["b." + n].grad))

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "block":
    block_debug()
    block_debug(128, 128, (2, 128, 4, 14, 14))


def mask_debug():
    spec = O.av_wrapper_spec(18)
    P = O.det_state(spec, "w")
    m = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128])
    m.load_state_dict({k: v.clone() for k, v in P.items()})
    m = m.to(dev).train()
    video = T(detgen.det_normalish("in:video", (2, 3, 8, 112, 112)))
    rec_cpu = {}
    orig_bn = O._bn
    def bn_rec(x, PP, prefix, train=True):
        y = orig_bn(x, PP, prefix, train)
        rec_cpu[prefix] = (x.detach(), y.detach())
        return y
    O._bn = bn_rec
    with torch.no_grad():
        pool_cpu = O.r2plus1d_forward(video, P, "video_model", 18, True)
    O._bn = orig_bn
    rec_gpu = {}
    from models.network_blocks import BatchNormCL
    for name, mod in m.video_model.named_modules():
        if isinstance(mod, BatchNormCL):
            mod.register_forward_hook(lambda md, inp, out, name=name: rec_gpu.__setitem__("video_model." + name, (inp[0].detach(), out.detach())))
    with torch.no_grad():
        pool_gpu = m.video_model(video.to(dev))
    for k in rec_cpu:
        xc, yc = rec_cpu[k]
        xg, yg = rec_gpu[k]
        xg = xg.permute(0, 4, 1, 2, 3).cpu(); yg = yg.permute(0, 4, 1, 2, 3).cpu()
        ypre = F.batch_norm(xc.double(), None, None, P[k + ".weight"].double(), P[k + ".bias"].double(), True)
        mism = ((yg > 0) != (F.relu(yc) > 0))
        print(f"{k:40s} xerr {float((xg-xc).abs().max()/xc.abs().max()):.2e} mask mismatches {int(mism.sum()):4d}/{mism.numel()}"
              f"  min|ypre| at mism {float(ypre.abs()[mism].min()) if mism.any() else -1:.2e}  frac |ypre|<1e-5: {float((ypre.abs()<1e-5).double().mean()):.2e}")

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "mask":
    mask_debug()
