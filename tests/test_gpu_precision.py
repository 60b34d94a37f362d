"""The arithmetic the bench line names — `f32 (bf16x3 MFMA)`: every fp32 product assembled from SIX bf16 matrix
instructions on operands split into three bf16 terms — pinned so that a kernel with fewer terms
fails.  The reference's arithmetic is fp32 `nn.Conv3d` (models/network_blocks.py:35-49, models/video.py:20).

DESIGN.md 8e's table (conv2x temporal, K = 192): six products rms 1.9e-7 of rms(y), the fp32 instruction 2.3e-7, THREE
products 4.2e-6, plain bf16 2.3e-3.  The per-kernel parity tests (tests/test_gpu_ops.py) bound max|err| by 2e-5 of max|y|,
which a three-product kernel passes by an order of magnitude; here the error is measured as rms against float64, relative
to rms(output), and must stay (a) below 6e-7 — a third of an order above six products, seven times below three — and
(b) within 3x of what a float32 `F.conv3d` on the CPU makes of the same data (the reference's own arithmetic; measured:
0.3x-2.1x — oneDNN's blocked fp32 accumulation on the host is itself more accurate than a straight fma chain for short
contractions: 1.3e-7 at K = 192-384 where the six-product kernels give 1.9-2.7e-7 and the fp32 matrix instruction 2.3e-7;
for the long contractions of the weight gradients the device is 2-6x MORE accurate than the host).  A three-product
kernel is 15-30x the host's error."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import detgen

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def cl(x):
    return x.permute(0, 2, 3, 4, 1).contiguous()


def ncdhw(x):
    return x.permute(0, 4, 1, 2, 3)


def rms_rel(got, want64):
    want64 = want64.double().cpu()
    return float(((got.double().cpu() - want64) ** 2).mean().sqrt() / ((want64 ** 2).mean().sqrt() + 1e-300))


# name, Cin, Cout, k, stride, pad, (B,T,H,W), channel_first, wino2 forced, kernels expected in the launch log
CASES = [
    # both stems' split-bf16 kernels: forward and weight gradient of models/video.py:20 at two clips
    ("stem", 3, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3), (2, 8, 112, 112), True, False,
     ("stem_fwd3p_kernel", "stem_wgrad3_kernel")),
    # 128 x 64 tile, weights pre-split (avid_wt_desc mode 5 / 6), input rows split in registers; wgrad_tab_kernel<1,3>
    ("pk_128x64_temporal", 64, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (3, 7, 40, 41), False, False,
     ("igemm_pk_kernel<4,1,1,2,0>", "igemm_pk_kernel<4,1,1,2,1>", "wgrad_tab_kernel")),
    # 128 x 128 tile, weights pre-split, as four waves of 32 x 128 (every input fragment split once); wgrad_tab_kernel<2,2>
    # (conv3x temporal at 32 clips)
    ("pk_128x128_temporal", 128, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), (32, 4, 14, 14), False, False,
     ("igemm_pk_kernel<4,1,1,4,0>", "igemm_pk_kernel<4,1,1,4,1>", "wgrad_tab_kernel<2,2>")),
    # 128 x 128 tile with a nine-way K-split + reduce (conv4x spatial off the Winograd path: 1568 pixels)
    ("pk_128x128_ksplit", 256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (16, 2, 7, 7), False, False,
     ("igemm_pk_kernel<4,1,1,4,0>", "igemm_pk_kernel<4,1,1,4,1>", "wgrad_tab_kernel<2,2>")),
    # the longest contraction of the network (conv5x spatial, K = 4608, sixteen-way K-split + reduce)
    ("pk_K4608", 512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1), (16, 1, 4, 4), False, False,
     ("igemm_pk_kernel<4,1,1,2,0>", "igemm_pk_kernel<4,1,1,2,1>", "wgrad_tab_kernel<2,2>")),
    # strided input gradients through the stride-parity classes, both tiles
    ("pk_strided_64", 64, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1), (4, 4, 28, 28), False, False,
     ("igemm_pk_kernel<4,1,1,2,0>", "igemm_pk_kernel<4,1,1,2,1>s2", "wgrad_tab_kernel<2,2>")),
    ("pk_strided_128", 128, 256, (1, 3, 3), (1, 2, 2), (0, 1, 1), (16, 2, 14, 14), False, False,
     ("igemm_pk_kernel<4,1,1,2,0>", "igemm_pk_kernel<4,1,1,2,1>s2", "wgrad_tab_kernel<2,2>")),
    # conv2x's temporal layers: tconv64_kernel (taps staged once, weights resident in LDS), forward and input gradient, and
    # twgrad64_kernel (the taps share their split fragments), weight gradient
    ("tconv64", 64, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (5, 8, 27, 29), False, False,
     ("tconv64_kernel<0>", "tconv64_kernel<1>", "twgrad64_kernel")),
    # Winograd F(2x2,3x3) with split-bf16 products (wino2_kernel, forward and input gradient)
    ("wino2_64", 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (4, 8, 48, 48), False, True, ("wino2p_kernel",)),
    ("wino2_128", 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), (9, 4, 27, 29), False, True, ("wino2p_kernel",)),
]
BAR = 6e-7          # rms(err) / rms(output) against float64


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_split_bf16_kernels_are_fp32_accurate(case, gpu_device, kernel_log):
    from avid_hip import lib, ops
    name, cin, cout, k, stride, pad, (B, Ti, Hi, Wi), channel_first, wino2, kernels = case
    x = T(detgen.det_normalish(f"prec:{name}:x", (B, cin, Ti, Hi, Wi)))
    w = T(detgen.det_param(f"prec:{name}:w.weight", (cout, cin) + k))
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv3d(xr, wr, stride=stride, padding=pad)
    gy = T(detgen.det_uniform(f"prec:{name}:gy", tuple(yr.shape)))
    (yr * gy.double()).sum().backward()
    # the reference's arithmetic on the same data: float32 conv3d on the host
    xf, wf = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yf = F.conv3d(xf, wf, stride=stride, padding=pad)
    (yf * gy).sum().backward()

    if wino2:
        ops.wino2_configure(0)
    if name == "tconv64":
        lib.raw("avid_tconv_configure")(2)
    try:
        xd = (x if channel_first else cl(x)).to(gpu_device).requires_grad_(not channel_first)
        wd = ops.make_weight(cout, cin, *k)
        wd.copy_(w)
        wd = wd.to(gpu_device).requires_grad_(True)
        with kernel_log() as log:
            y = ops.conv_cl(xd, wd, stride, pad, channel_first=channel_first)
            y.backward(cl(gy).to(gpu_device))
    finally:
        if wino2:
            ops.wino2_configure(-1)
        lib.raw("avid_tconv_configure")(-1)
    for kn in kernels:
        assert log.launches(kn) >= 1, (kn, sorted(log.report))
    got = {"y": (ncdhw(y.detach()), yr.detach(), yf.detach()), "dw": (wd.grad, wr.grad, wf.grad)}
    if not channel_first:
        got["dx"] = (ncdhw(xd.grad), xr.grad, xf.grad)
    report = {}
    for what, (dev, ref64, ref32) in got.items():
        e_dev, e_f32 = rms_rel(dev, ref64), rms_rel(ref32, ref64)
        report[what] = (e_dev, e_f32)
    print(f"\n[precision] {name}: " + ", ".join(f"{q} {a:.2e} (float32 conv3d {b:.2e})" for q, (a, b) in report.items()))
    for what, (e_dev, e_f32) in report.items():
        assert e_dev <= BAR, (name, what, e_dev)
        assert e_dev <= 3.0 * e_f32, (name, what, e_dev, e_f32)


def test_grouped_weight_gradient_is_fp32_accurate(gpu_device, kernel_log):
    """wgrad_group_kernel (one launch over a table of layers; both fragments split in registers): same bars."""
    from avid_hip import lib, ops
    layers = [(128, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), (8, 4, 14, 14)),
              (256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (8, 2, 7, 7)),
              (512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1), (16, 1, 4, 4)),
              (128, 256, (1, 1, 1), (2, 2, 2), (0, 0, 0), (8, 4, 14, 14))]
    items = (lib.WgradItem * len(layers))()
    keep, refs, outs = [], [], []
    for i, (cin, cout, k, stride, pad, (B, Ti, Hi, Wi)) in enumerate(layers):
        x = T(detgen.det_normalish(f"precg:{i}:x", (B, cin, Ti, Hi, Wi)))
        w = T(detgen.det_param(f"precg:{i}:w.weight", (cout, cin) + k))
        wr = w.double().requires_grad_(True)
        yr = F.conv3d(x.double(), wr, stride=stride, padding=pad)
        gy = T(detgen.det_uniform(f"precg:{i}:gy", tuple(yr.shape)))
        (yr * gy.double()).sum().backward()
        wf = w.clone().requires_grad_(True)
        (F.conv3d(x, wf, stride=stride, padding=pad) * gy).sum().backward()
        refs.append((wr.grad, wf.grad))
        xd, gyd = cl(x).to(gpu_device), cl(gy).to(gpu_device)
        d = ops._desc_cached((B, Ti, Hi, Wi), cin, cout, k, stride, pad, False)[0]
        assert d.groupable
        dw = ops.make_weight(cout, cin, *k).to(gpu_device).fill_(float("nan"))
        items[i].d = d
        items[i].x, items[i].dy, items[i].dw = xd.data_ptr(), gyd.data_ptr(), dw.data_ptr()
        keep += [xd, gyd]
        outs.append(dw)
    nb = lib.raw("avid_conv_wgrad_group_workspace_bytes")(len(layers), items)
    ws = torch.empty(max(int(nb), 16), dtype=torch.uint8, device=gpu_device)
    with kernel_log() as log:
        lib.call("avid_conv_wgrad_group", len(layers), items, ops._p(ws), ws.numel(), ops._stream())
    assert log.launches("wgrad_group_kernel") == 1
    for i, (o, (r64, r32)) in enumerate(zip(outs, refs)):
        e_dev, e_f32 = rms_rel(o, r64), rms_rel(r32, r64)
        print(f"\n[precision] wgrad_group layer {i}: dw {e_dev:.2e} (float32 conv3d {e_f32:.2e})")
        assert e_dev <= BAR and e_dev <= 3.0 * e_f32, (i, e_dev, e_f32)


def test_non_finite_operand_gives_nan_and_nothing_else_changes(gpu_device, monkeypatch):
    """The documented change in kind: inf = hi + mid + lo has mid = bf16(inf - inf) = NaN, so every output the
    non-finite input element reaches is NaN where the fp32 instruction delivers +-inf; outputs it does not reach are
    bit-identical to the same convolution with a finite value in its place."""
    from avid_hip import ops
    cin = cout = 64
    k, stride, pad, (B, Ti, Hi, Wi) = (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 5, 12, 13)
    x = T(detgen.det_normalish("prec:inf:x", (B, Ti, Hi, Wi, cin)))
    w = T(detgen.det_param("prec:inf:w.weight", (cout, cin) + k))
    wd = ops.make_weight(cout, cin, *k)
    wd.copy_(w)
    wd = wd.to(gpu_device)
    d = ops._desc_cached((B, Ti, Hi, Wi), cin, cout, k, stride, pad, False)[0]
    assert d.split_fwd
    b, t, hh, ww, c = 1, 2, 5, 7, 9
    x_fin, x_inf, x_big = x.clone(), x.clone(), x.clone()
    x_fin[b, t, hh, ww, c] = 0.0
    x_inf[b, t, hh, ww, c] = float("inf")
    x_big[b, t, hh, ww, c] = 3.4e38             # finite, above the largest bf16 + half an ulp (3.3962e38): hi rounds to inf
    y_fin = ops.conv_cl(x_fin.to(gpu_device), wd, stride, pad).cpu()
    reached = torch.zeros(B, Ti, Hi, Wi, dtype=torch.bool)
    reached[b, max(t - 1, 0):t + 2, hh, ww] = True          # the (3,1,1) taps that read frame t at this position
    for xin in (x_inf, x_big):
        y = ops.conv_cl(xin.to(gpu_device), wd, stride, pad).cpu()
        assert bool(torch.isnan(y[reached]).all())
        assert torch.equal(y[~reached], y_fin[~reached])
    # the fp32 matrix instruction on the same layer (no pre-split table): inf stays inf
    monkeypatch.setattr(ops, "_split_for", lambda w_, mode: None)
    y32 = ops.conv_cl(x_inf.to(gpu_device), wd, stride, pad).cpu()
    assert bool(torch.isinf(y32[reached]).all())
    assert torch.equal(y32[~reached], ops.conv_cl(x_fin.to(gpu_device), wd, stride, pad).cpu()[~reached])


def test_uses_split_agrees_with_the_kernel_that_ran(gpu_device, kernel_log):
    """avid_conv_uses_split (what ops / plan.py consult to hand a layer its pre-split weight table) against what the
    library launched: the pre-split form of igemm_pk_kernel exactly where it says so, the kernel name
    avid_conv_kernel_name predicts, and stem_fwd3_kernel for the video stem."""
    from avid_hip import lib, ops
    count = lib.raw("avid_debug_presplit_launches")
    cases = [  # cin, cout, k, stride, pad, shape, channel_first
        (64, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (3, 7, 40, 41), False),
        (128, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), (9, 2, 45, 47), False),
        (64, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), (4, 4, 14, 14), False),
        (256, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 1, 4, 4), False),
        (64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (4, 8, 48, 48), False),
        (3, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3), (2, 8, 112, 112), True),
    ]
    seen_split = seen_plain = 0
    for cin, cout, k, stride, pad, (B, Ti, Hi, Wi), cf in cases:
        d = ops._desc_cached((B, Ti, Hi, Wi), cin, cout, k, stride, pad, cf)[0]
        x = T(detgen.det_normalish(f"prec:us:{cin}:{cout}", (B, cin, Ti, Hi, Wi)))
        xd = (x if cf else cl(x)).to(gpu_device).requires_grad_(not cf)
        wd = ops.make_weight(cout, cin, *k)
        wd.copy_(T(detgen.det_param(f"prec:us:{cin}:{cout}:w.weight", (cout, cin) + k)))
        wd = wd.to(gpu_device).requires_grad_(True)
        buf = C.create_string_buffer(128)
        names = []
        for which in (0, 1):
            lib.call("avid_conv_kernel_name", C.byref(d), which, buf, 128)
            names.append(buf.value.decode())
        before = count()
        with kernel_log() as log:
            y = ops.conv_cl(xd, wd, stride, pad, channel_first=cf)
        mid = count()
        with kernel_log() as log_b:
            y.backward(torch.ones_like(y))
        after = count()
        assert mid - before == (1 if d.split_fwd else 0), (cin, cout, k, names)
        assert after - mid == (1 if d.split_dgrad else 0), (cin, cout, k, names)
        seen_split += int(d.split_fwd) + int(d.split_dgrad)
        seen_plain += int(not d.split_fwd) + int(not d.split_dgrad and not cf)
        fwd_kernel = names[0].split(" ")[0]
        assert log.launches(fwd_kernel) == 1, (names, sorted(log.report))
        if cf:
            assert fwd_kernel.startswith("stem_fwd3p_kernel") and d.split_fwd is False      # (its own split: no table)
        elif not d.wino_dgrad and "sub-sampled" not in names[1]:
            assert log_b.launches(names[1].split(" ")[0]) == 1, (names, sorted(log_b.report))
    assert seen_split >= 4 and seen_plain >= 3
