"""Per-kernel parity on a real MI355X: every HIP op (called through the C-ABI via avid_hip.ops)
against the same op computed on the CPU by torch in float64 (encoder ops) or by the oracle
(criterion ops).  Tolerances are stated per test; integer / index work is bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import avid_oracle as O
from oracle import detgen

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def cl(x):      # NCDHW -> channels-last contiguous
    return x.permute(0, 2, 3, 4, 1).contiguous()


def ncdhw(x):   # channels-last -> NCDHW (logical)
    return x.permute(0, 4, 1, 2, 3)


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


CONV_CASES = [
    # name, Cin, Cout, k, stride, pad, (B,T,H,W)
    ("spt_s1", 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 3, 9, 11)),
    ("spt_s2", 64, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1), (2, 4, 10, 12)),
    ("spt_s2_odd", 64, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 2, 7, 9)),
    ("tmp_s1", 128, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 4, 5, 6)),
    ("tmp_s2", 128, 128, (3, 1, 1), (2, 1, 1), (1, 0, 0), (2, 4, 5, 6)),
    ("tmp_s2_odd", 64, 64, (3, 1, 1), (2, 1, 1), (1, 0, 0), (2, 5, 3, 4)),
    ("res_s2", 64, 128, (1, 1, 1), (2, 2, 2), (0, 0, 0), (2, 4, 10, 12)),
    ("late_small_m", 256, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 1, 4, 4)),
    ("late_64x64", 256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (3, 2, 14, 14)),
    ("big_128x64", 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (4, 8, 48, 48)),
    ("big_128x128", 64, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), (4, 8, 48, 48)),
    # large temporal layers: an odd number of frames, T = 2
    ("tmp_odd_T", 64, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (3, 7, 40, 41)),
    ("tmp_T2_128", 128, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), (9, 2, 45, 47)),
    # persistent kernel: whole rounds + split-K remainder with a ragged last tile; 27 taps crossing batch items
    ("big_ragged_333", 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), (5, 7, 45, 47)),
    # persistent kernel alone, last round 80 % full and ragged (run unbalanced)
    ("big_unbalanced", 64, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), (7, 8, 45, 47)),
    # dead temporal taps (conv5x of R(2+1)D-18): T = 1 with a (3,1,1) kernel -> only the centre tap is live;
    # T = 2 -> 1 at stride 2 -> the first tap is dead; (3,3,3) on T = 1; the trimmed layers run as narrower kernels
    ("dead_taps_T1", 512, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0), (16, 1, 4, 4)),
    ("dead_taps_T2_s2", 128, 128, (3, 1, 1), (2, 1, 1), (1, 0, 0), (8, 2, 4, 4)),
    ("dead_taps_333_T1", 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), (4, 1, 6, 7)),
    # Winograd F(2x2,3x3) with Cr = Cn = 128 (two 64-column blocks x four reduction chunks): the benchmark's conv3x
    # layer (25088 pixels at batch 32; models/network_blocks.py:35,40) and an odd-extent sibling (ragged tiles)
    ("wino_128x128", 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), (32, 4, 14, 14)),
    ("wino_128x128_odd", 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), (9, 4, 27, 29)),
    # 256 channels (conv4x at the benchmark batch: 6272 pixels): four column blocks x eight reduction chunks, fewer tile
    # blocks than workgroup slots; and the smallest 64-channel layer the dispatch rule admits (audio block 1: 16000 pixels)
    ("wino_256x256", 256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (8, 2, 21, 19)),
    ("wino_64x64_audio", 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (64, 1, 10, 25)),
]
WINO_CASES = {"big_128x64", "big_unbalanced", "wino_128x128", "wino_128x128_odd", "wino_256x256", "wino_64x64_audio"}


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(case, gpu_device, kernel_log, wino_variant):
    """fp32 MFMA implicit GEMM vs float64 F.conv3d.  Tolerance: 2e-5 of the output scale
    (fp32 products/accumulation over K <= 2304; the reference itself is fp32)."""
    from avid_hip import ops
    name, cin, cout, k, stride, pad, (B, Ti, Hi, Wi) = case
    x = T(detgen.det_normalish(f"conv:{name}:x", (B, cin, Ti, Hi, Wi)))
    w = T(detgen.det_param(f"conv:{name}:w.weight", (cout, cin) + k))
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv3d(xr, wr, stride=stride, padding=pad)
    gy = T(detgen.det_uniform(f"conv:{name}:gy", tuple(yr.shape)))
    (yr * gy.double()).sum().backward()

    xd = cl(x).to(gpu_device).requires_grad_(True)
    wd = ops.make_weight(cout, cin, *k)
    wd.copy_(w)
    wd = wd.to(gpu_device).requires_grad_(True)
    assert ops.weight_layout_ok(wd)
    with kernel_log() as log:
        y = ops.conv_cl(xd, wd, stride, pad)
        y.backward(cl(gy).to(gpu_device))
    assert relerr(ncdhw(y.detach()), yr.detach()) < 2e-5
    assert relerr(ncdhw(xd.grad), xr.grad) < 2e-5
    assert relerr(wd.grad, wr.grad) < 5e-5
    assert wd.grad.stride() == wd.stride()
    if wino_variant == "wino2":
        assert log.launches("wino_kernel") == 0
    if name in WINO_CASES:          # forward and input gradient really took the Winograd kernel
        assert log.launches("wino_kernel") + log.launches("wino2_kernel") + log.launches("wino2p_kernel") == 2 and log.launches("wino_wgrad_kernel") == 1, log.report.keys()
    else:
        assert log.launches("wino_kernel") + log.launches("wino2_kernel") + log.launches("wino2p_kernel") == 0 and log.launches("wino_wgrad_kernel") == 0


@pytest.mark.parametrize("mode", [5, 6])
@pytest.mark.parametrize("cout,cin,k", [(64, 64, (1, 3, 3)), (192, 128, (3, 1, 1)), (64, 256, (1, 1, 1))])
def test_presplit_weight_tables_hold_the_weights(cout, cin, k, mode, gpu_device):
    """avid_wt_desc mode 5 / 6 (what igemm_pk_kernel's 128 x 64 tile reads instead of fp32 weights): three bf16 terms per
    weight, hi + mid + lo == w to 2^-24 of |w|, each at the fragment position the kernel reads it from —
    [tap][32-channel block][64-row block][row half][k-step of 16][term][lane = 32 * (k / 8 % 2) + row % 32][k % 8]."""
    from avid_hip import ops
    w = ops.make_weight(cout, cin, *k)
    w.copy_(T(detgen.det_param(f"split:{cout}:{cin}:{k}", (cout, cin) + k)))
    w = w.to(gpu_device)
    planes = ops._split_for(w, mode)
    torch.cuda.synchronize()
    taps = k[0] * k[1] * k[2]
    W = w.permute(0, 2, 3, 4, 1).reshape(cout, taps, cin).cpu().numpy()      # [row][tap][channel] of the forward operand
    if mode == 6:
        W = W.transpose(2, 1, 0)                                               # input gradient: rows = Cin, channels = Cout
    N, _, Cc = W.shape
    raw = planes.cpu().numpy().view(np.uint16).astype(np.uint32) << 16
    terms = raw.view(np.float32).reshape(taps, Cc // 32, N // 64, 2, 2, 3, 2, 32, 8).astype(np.float64)
    got = terms.sum(axis=5)                                                    # [tap][cb][nt][j][st][h][l31][e]
    want = W.reshape(N // 64, 2, 32, taps, Cc // 32, 2, 2, 8).transpose(3, 4, 0, 1, 5, 6, 2, 7)   # nt j l31 tap cb st h e ->
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 2.0 ** -23 * np.abs(W).max()
    hi = terms[:, :, :, :, :, 0]
    assert np.abs(hi - want).max() <= 2.0 ** -8 * np.abs(W).max()             # hi alone is the bf16 rounding of w


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[0] in ("spt_s1", "tmp_s2_odd", "big_ragged_333", "dead_taps_333_T1",
                                                                    "tmp_odd_T")], ids=lambda c: c[0])
def test_conv_presplit_weights_against_the_fp32_instruction(case, gpu_device, monkeypatch):
    """The same layer with its weights pre-split (six bf16 matrix instructions per product tile) and without (u = NULL:
    the fp32 matrix instruction): both within the usual 2e-5 of float64, and within 1e-5 of each other."""
    from avid_hip import ops
    name, cin, cout, k, stride, pad, (B, Ti, Hi, Wi) = case
    x = T(detgen.det_normalish(f"conv:{name}:x", (B, cin, Ti, Hi, Wi)))
    w = T(detgen.det_param(f"conv:{name}:w.weight", (cout, cin) + k))
    yr = F.conv3d(x.double().requires_grad_(True), w.double(), stride=stride, padding=pad)
    gy = T(detgen.det_uniform(f"conv:{name}:gy", tuple(yr.shape)))
    d = ops._desc_cached((B, Ti, Hi, Wi), cin, cout, k, stride, pad, False)[0]
    assert d.split_fwd and d.split_dgrad, "case does not run on the 128 x 64 tile"
    outs = []
    for presplit in (True, False):
        if not presplit:
            monkeypatch.setattr(ops, "_split_for", lambda w_, mode: None)
        xd = cl(x).to(gpu_device).requires_grad_(True)
        wd = ops.make_weight(cout, cin, *k)
        wd.copy_(w)
        wd = wd.to(gpu_device).requires_grad_(True)
        y = ops.conv_cl(xd, wd, stride, pad)
        y.backward(cl(gy).to(gpu_device))
        outs.append((ncdhw(y.detach()).cpu(), ncdhw(xd.grad).cpu()))
    xr = x.double().requires_grad_(True)
    yr = F.conv3d(xr, w.double(), stride=stride, padding=pad)
    (yr * gy.double()).sum().backward()
    for y, dx in outs:
        assert relerr(y, yr.detach()) < 2e-5 and relerr(dx, xr.grad) < 2e-5
    assert relerr(outs[0][0], outs[1][0].double()) < 1e-5 and relerr(outs[0][1], outs[1][1].double()) < 1e-5
    assert not torch.equal(outs[0][0], outs[1][0])        # (two different instruction sequences really ran)


@pytest.mark.parametrize("shape", [(2, 3, 9, 11), (4, 8, 48, 48), (5, 7, 45, 47), (64, 1, 4, 4)])
@pytest.mark.parametrize("cout", [64, 128, 256])
@pytest.mark.parametrize("cin", [64, 128])
def test_conv_bn_partials(shape, cout, cin, gpu_device, wino_variant):
    """BatchNorm partial sums written by the conv epilogue / the split-K reduce: their column totals must be
    the column sums and sums of squares of the conv output (fp32 partials, 1e-5 of the scale), for whole
    tiles, ragged tails and K-split layers alike; with a fused residual add the statistics are of the sum."""
    from avid_hip import ops
    B, Ti, Hi, Wi = shape
    x = T(detgen.det_normalish(f"cbp:{shape}:{cin}:x", (B, Ti, Hi, Wi, cin))).to(gpu_device)
    w = ops.make_weight(cout, cin, 1, 3, 3)
    w.copy_(T(detgen.det_param(f"cbp:{cout}:{cin}:w.weight", (cout, cin, 1, 3, 3))))
    w = w.to(gpu_device)
    add = T(detgen.det_uniform(f"cbp:{shape}:{cout}:add", (B, Ti, Hi, Wi, cout))).to(gpu_device)
    for addend in (None, add):
        y, part = ops.conv_cl(x, w, (1, 1, 1), (0, 1, 1), addend=addend, bn_stats=True)
        y_plain = ops.conv_cl(x, w, (1, 1, 1), (0, 1, 1), addend=addend)
        assert torch.equal(y, y_plain)                         # the statistics do not perturb the output
        if part.numel() == 0:
            pytest.skip("this configuration (AVID_PK=0) cannot produce conv-side BatchNorm partials")
        assert part.dim() == 3 and part.shape[1:] == (2, cout)
        yd = y.double().reshape(-1, cout)
        s, q = part[:, 0].double().sum(0), part[:, 1].double().sum(0)
        assert relerr(s, yd.sum(0)) < 1e-5 * max(1.0, float(yd.abs().sum(0).max() / (yd.sum(0).abs().max() + 1e-30)))
        assert relerr(q, (yd * yd).sum(0)) < 1e-5


@pytest.mark.parametrize("shape,cmid,cout,k,pad,stride", [
    ((2, 4, 12, 12), 64, 64, (1, 3, 3), (0, 1, 1), (1, 1, 1)),        # one partial tile
    ((7, 8, 45, 47), 64, 128, (3, 1, 1), (1, 0, 0), (1, 1, 1)),       # whole rounds + a K-split tail (reduce kernel)
    ((4, 3, 17, 19), 128, 128, (1, 3, 3), (0, 1, 1), (1, 1, 1)),      # 128-wide tiles, ragged
    ((8, 8, 28, 28), 64, 128, (1, 3, 3), (0, 1, 1), (1, 2, 2)),       # strided consumer, tiles written directly
    ((3, 4, 13, 15), 128, 256, (1, 3, 3), (0, 1, 1), (1, 2, 2)),      # strided, every tile K-split (reduce kernel)
    ((3, 5, 9, 9), 64, 64, (3, 1, 1), (1, 0, 0), (2, 1, 1)),          # temporal stride, odd extent
    ((6, 8, 27, 29), 64, 64, (1, 3, 3), (0, 1, 1), (1, 1, 1)),        # Winograd input gradient (>= 32768 pixels), odd extents
    ((5, 4, 45, 47), 128, 128, (1, 3, 3), (0, 1, 1), (1, 1, 1)),      # Winograd, two 64-column blocks, 4 reduction chunks
    ((8, 2, 21, 19), 256, 256, (1, 3, 3), (0, 1, 1), (1, 1, 1)),      # Winograd at 256 channels (conv4x): 4 column blocks, 8 chunks
    ((3, 7, 40, 41), 64, 64, (3, 1, 1), (1, 0, 0), (1, 1, 1)),        # large temporal layer, odd frame count
])
def test_bn_backward_partials_from_dgrad(shape, cmid, cout, k, pad, stride, gpu_device, wino_variant):
    """conv1 -> BN+ReLU -> conv2 [+ tap]: with ops.BnSource the BatchNorm's backward partial sums come out of
    conv2's input-gradient kernel (epilogue / K-split reduce) instead of the BatchNorm's own pass over dy and x.
    Gradients vs the same chain without the hand-over: 2e-5 of the gradient scale (fp32 partial sums in a
    different order); the consumer conv's output gradient dx is bit-identical (the sums only ride along)."""
    from avid_hip import ops
    B, Ti, Hi, Wi = shape
    x = T(detgen.det_normalish(f"bnb:{shape}:x", (B, Ti, Hi, Wi, 64))).to(gpu_device)
    w1 = ops.make_weight(cmid, 64, 1, 3, 3); w1.copy_(T(detgen.det_param(f"bnb:{cmid}:w1.weight", (cmid, 64, 1, 3, 3))))
    w2 = ops.make_weight(cout, cmid, *k); w2.copy_(T(detgen.det_param(f"bnb:{cout}:{k}:w2.weight", (cout, cmid) + k)))
    w1, w2 = w1.to(gpu_device), w2.to(gpu_device)
    gam = (T(detgen.det_uniform(f"bnb:{cmid}:g", (cmid,))) + 1.5).to(gpu_device)
    bet = T(detgen.det_uniform(f"bnb:{cmid}:b", (cmid,))).to(gpu_device)
    gy = None
    res = {}
    for fused in (False, True):
        for tap in (False, True):
            xx = x.clone().requires_grad_(True)
            g_, b_ = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
            rm, rv = torch.zeros(cmid, device=gpu_device), torch.ones(cmid, device=gpu_device)
            y1 = ops.conv_cl(xx, w1, (1, 1, 1), (0, 1, 1))
            src = ops.BnSource(None, None, True) if fused else None
            h = ops.batch_norm_cl(y1, g_, b_, rm, rv, True, relu=True, src=src)
            out = ops.conv_cl(h, w2, stride, pad, tap=tap, bn_src=src)
            y2, alias = (out[0], out[-1]) if tap else (out, None)
            if gy is None:
                gy = T(detgen.det_uniform(f"bnb:{shape}:{cout}:gy", tuple(y2.shape))).to(gpu_device)
            loss = (y2 * gy).sum()
            if tap:
                loss = loss + (alias * alias).sum() * 0.25          # a second consumer through the tap
            loss.backward()
            if fused:
                assert src.partials is None                          # consumed by the BatchNorm's backward
            res[(fused, tap)] = (xx.grad.clone(), g_.grad.clone(), b_.grad.clone())
    for tap in (False, True):
        for a, b in zip(res[(False, tap)], res[(True, tap)]):
            assert relerr(b, a) < 2e-5


GROUP_LAYERS = [
    # Cin, Cout, k, stride, pad, (B, T, H, W): the small layers of the model at a small batch + the shapes that take the
    # special paths inside a group: dead temporal taps (T = 1), strided, 1x1x1 residual, a linear layer, a K-split one
    (512, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0), (16, 1, 4, 4)),       # dead taps: only the centre tap is live
    (256, 512, (1, 3, 3), (1, 2, 2), (0, 1, 1), (8, 2, 7, 7)),
    (256, 512, (1, 1, 1), (2, 2, 2), (0, 0, 0), (8, 2, 7, 7)),
    (512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1), (8, 1, 4, 4)),
    (512, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), (40, 1, 1, 1)),       # a head's linear layer
    (128, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), (4, 4, 14, 14)),      # many pixel chunks: gets K-split inside the group
    (64, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1), (3, 4, 9, 11)),
    (128, 256, (3, 1, 1), (2, 1, 1), (1, 0, 0), (5, 4, 7, 7)),
    (256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (4, 2, 7, 7)),
    (128, 128, (3, 1, 1), (2, 1, 1), (1, 0, 0), (8, 2, 4, 4)),        # T = 2 -> 1: the first tap is dead
    (64, 128, (1, 1, 1), (2, 2, 2), (0, 0, 0), (3, 4, 10, 12)),
    (512, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), (64, 1, 1, 1)),
]


@pytest.mark.parametrize("count", [12, 5, 1])
def test_grouped_weight_gradients(count, gpu_device):
    """avid_conv_wgrad_group (one persistent launch over a table of layers + one grouped reduce) against float64
    conv3d weight gradients (5e-5 of each gradient's scale, as the per-layer kernel) — dead temporal taps come out as
    exact zeros — and against avid_conv_wgrad layer by layer (1e-5: same products, another summation order over pixel
    chunks); a second launch is bit-identical."""
    import ctypes as C
    from avid_hip import lib, ops
    layers = GROUP_LAYERS[:count]
    items = (lib.WgradItem * count)()
    keep, refs, singles, outs = [], [], [], []
    for i, (cin, cout, k, stride, pad, (B, Ti, Hi, Wi)) in enumerate(layers):
        x = T(detgen.det_normalish(f"grp:{i}:x", (B, cin, Ti, Hi, Wi)))
        w = T(detgen.det_param(f"grp:{i}:w.weight", (cout, cin) + k))
        wr = w.double().requires_grad_(True)
        yr = F.conv3d(x.double(), wr, stride=stride, padding=pad)
        gy = T(detgen.det_uniform(f"grp:{i}:gy", tuple(yr.shape)))
        (yr * gy.double()).sum().backward()
        refs.append(wr.grad)
        xd, gyd = cl(x).to(gpu_device), cl(gy).to(gpu_device)
        d, _, _, nbw, _ = ops._desc_cached((B, Ti, Hi, Wi), cin, cout, k, stride, pad, False)
        assert d.groupable
        dw = ops.make_weight(cout, cin, *k).to(gpu_device).fill_(float("nan"))     # every element must be written
        one = ops.make_weight(cout, cin, *k).to(gpu_device)
        ws = ops.workspace(gpu_device, nbw)
        lib.call("avid_conv_wgrad", C.byref(d), ops._p(xd), ops._p(gyd), ops._p(one), ops._p(ws), ws.numel(), ops._stream())
        singles.append(one)
        items[i].d = d
        items[i].x, items[i].dy, items[i].dw = xd.data_ptr(), gyd.data_ptr(), dw.data_ptr()
        keep += [xd, gyd]
        outs.append(dw)
    nb = lib.raw("avid_conv_wgrad_group_workspace_bytes")(count, items)
    ws = torch.empty(max(int(nb), 16), dtype=torch.uint8, device=gpu_device)
    lib.call("avid_conv_wgrad_group", count, items, ops._p(ws), ws.numel(), ops._stream())
    first = [o.clone() for o in outs]
    for o in outs:
        o.fill_(float("nan"))
    lib.call("avid_conv_wgrad_group", count, items, ops._p(ws), ws.numel(), ops._stream())
    for i, (o, ref, one) in enumerate(zip(outs, refs, singles)):
        assert torch.equal(o, first[i])
        assert relerr(o, ref) < 5e-5, (i, relerr(o, ref))
        assert relerr(o, one) < 1e-5, (i, relerr(o, one))
        dead = (ref == 0).all(0).all(0).all(-1).all(-1) if ref.dim() == 5 else None     # temporal taps that only meet padding
        if dead is not None and bool(dead.any()):
            assert bool((o.cpu()[:, :, dead] == 0).all())


BIG_GROUP = [(128, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), (64, 4, 14, 14))] * 6 + \
            [(128, 256, (1, 3, 3), (1, 2, 2), (0, 1, 1), (64, 4, 14, 14))] * 6


@pytest.mark.parametrize("layers", [GROUP_LAYERS[:12], BIG_GROUP], ids=["small_layers", "conv3x_at_64_clips"])
def test_grouped_weight_gradients_presplit_is_bit_identical(layers, gpu_device):
    """wgrad_group_kernel with the fragments split once at the LDS write (wgrad_pre_body: k-step stages, three bf16 planes,
    permuted fragment slots) against the form that gathers and splits per use: same split, same products, same order — every
    layer of the twelve-layer table (strided, temporal, 1x1x1, dead taps, ragged pixel counts), slabs and direct writes; and
    twelve conv3x-sized layers at 64 clips, whose items are longer than one fill of the row table (64 chunks)."""
    import ctypes as C
    from avid_hip import lib, ops
    count = len(layers)
    items = (lib.WgradItem * count)()
    keep, outs = [], []
    for i, (cin, cout, k, stride, pad, (B, Ti, Hi, Wi)) in enumerate(layers):
        x = T(detgen.det_normalish(f"grp:{i}:x", (B, cin, Ti, Hi, Wi)))
        To, Ho, Wo = [(n + 2 * p_ - k_) // s_ + 1 for n, p_, k_, s_ in zip((Ti, Hi, Wi), pad, k, stride)]
        gy = T(detgen.det_uniform(f"grp:{i}:gy", (B, cout, To, Ho, Wo)))
        xd, gyd = cl(x).to(gpu_device), cl(gy).to(gpu_device)
        d, _, _, _, _ = ops._desc_cached((B, Ti, Hi, Wi), cin, cout, k, stride, pad, False)
        dw = ops.make_weight(cout, cin, *k).to(gpu_device)
        items[i].d = d
        items[i].x, items[i].dy, items[i].dw = xd.data_ptr(), gyd.data_ptr(), dw.data_ptr()
        keep += [xd, gyd]
        outs.append(dw)
    nb = lib.raw("avid_conv_wgrad_group_workspace_bytes")(count, items)
    ws = torch.empty(max(int(nb), 16), dtype=torch.uint8, device=gpu_device)
    pre = lib.raw("avid_wgrad_pre_configure")
    res = {}
    try:
        for on in (1, 0):
            pre(on)
            for o in outs:
                o.fill_(float("nan"))
            lib.call("avid_conv_wgrad_group", count, items, ops._p(ws), ws.numel(), ops._stream())
            res[on] = [o.clone() for o in outs]
    finally:
        pre(-1)
    for i, (a, b) in enumerate(zip(res[1], res[0])):
        assert bool(torch.isfinite(a).all()), i
        assert torch.equal(a, b), (i, float((a - b).abs().max()))


def test_conv_transpose_detecting(gpu_device):
    """A = identity-like with ASYMMETRIC weights: catches a row<->col swap in the MFMA C-write."""
    from avid_hip import ops
    cin = cout = 64
    x = torch.zeros(1, cin, 1, 1, 70)
    for i in range(64):
        x[0, i, 0, 0, i] = 1.0                       # pixel i carries channel i
    w = torch.zeros(cout, cin, 1, 1, 1)
    for o in range(cout):
        for i in range(cin):
            w[o, i] = o * 100 + i                    # asymmetric
    wd = ops.make_weight(cout, cin, 1, 1, 1)
    wd.copy_(w)
    y = ops.conv_cl(cl(x).to(gpu_device), wd.to(gpu_device), (1, 1, 1), (0, 0, 0))
    yr = F.conv3d(x, w)
    assert torch.equal(ncdhw(y).cpu(), yr)


@pytest.mark.parametrize("which", ["video", "audio"])
def test_stem_conv(which, gpu_device):
    """The two stems read the reference's channel-first input directly (LDS-patch kernels, csrc/stem.hip)."""
    from avid_hip import ops
    if which == "video":
        cin, k, stride, pad, shp = 3, (3, 7, 7), (1, 2, 2), (1, 3, 3), (2, 3, 4, 20, 26)
    else:
        cin, k, stride, pad, shp = 1, (1, 7, 7), (1, 2, 2), (0, 3, 3), (3, 1, 1, 40, 100)
    x = T(detgen.det_normalish(f"stem:{which}:x", shp))
    w = T(detgen.det_param(f"stem:{which}:w.weight", (64, cin) + k))
    wr = w.double().requires_grad_(True)
    yr = F.conv3d(x.double(), wr, stride=stride, padding=pad)
    gy = T(detgen.det_uniform(f"stem:{which}:gy", tuple(yr.shape)))
    (yr * gy.double()).sum().backward()
    wd = ops.make_weight(64, cin, *k)
    wd.copy_(w)
    wd = wd.to(gpu_device).requires_grad_(True)
    y = ops.conv_cl(x.to(gpu_device), wd, stride, pad, channel_first=True)
    y.backward(cl(gy).to(gpu_device))
    assert relerr(ncdhw(y.detach()), yr.detach()) < 2e-5
    assert relerr(wd.grad, wr.grad) < 5e-5


@pytest.mark.parametrize("shp", [(2, 3, 8, 112, 112), (5, 3, 2, 38, 44), (2, 3, 4, 24, 28), (1, 3, 1, 112, 112)],
                         ids=["clips_112", "ragged_tiles", "small", "one_frame"])
def test_stem_fwd_presplit_patch(shp, gpu_device, kernel_log):
    """stem_fwd3p_kernel (the patch split once at commit time into three bf16 planes, a lane's fragment = eight consecutive
    columns of one row) against float64 and against stem_fwd3_kernel (fp32 patch, taps gathered and split per k-step): the same
    products per (row, tap) at other k positions of the matrix instruction — outputs agree to rounding, BatchNorm partial sums
    are the sums of what was written; temporal padding (one frame: two dead planes), ragged last tiles, both image borders."""
    from avid_hip import lib, ops
    k, stride, pad = (3, 7, 7), (1, 2, 2), (1, 3, 3)
    x = T(detgen.det_normalish(f"stemp:{shp}:x", shp))
    w = T(detgen.det_param(f"stemp:{shp}:w.weight", (64, 3) + k))
    yr = F.conv3d(x.double(), w.double(), stride=stride, padding=pad)
    wd = ops.make_weight(64, 3, *k)
    wd.copy_(w)
    xd, wd = x.to(gpu_device), wd.to(gpu_device)
    pre = lib.raw("avid_stem_fwd_pre_configure")
    res = {}
    try:
        for on in (1, 0):
            pre(on)
            with kernel_log() as log:
                y, part = ops.conv_cl(xd, wd, stride, pad, channel_first=True, bn_stats=True)
            # which form ran, from the launch log: the timers name the two kernels apart
            assert log.launches("stem_fwd3p_kernel") == (1 if on else 0) and log.launches("stem_fwd3_kernel") == (0 if on else 1), sorted(log.report)
            res[on] = (ncdhw(y).cpu(), part.clone())
    finally:
        pre(-1)
    for on in (1, 0):
        y, part = res[on]
        e = float((y.double() - yr).pow(2).mean().sqrt() / yr.pow(2).mean().sqrt())
        assert e <= 6e-7, (on, e)                                    # the precision tests' bar (a three-product kernel: 4e-6)
        yd = y.double().permute(0, 2, 3, 4, 1).reshape(-1, 64)
        assert relerr(part[:, 1].double().sum(0).cpu(), (yd * yd).sum(0)) < 1e-5
    assert relerr(res[1][0], res[0][0]) < 2e-6


@pytest.mark.parametrize("which", ["video", "audio", "video_ragged"])
def test_stem_bn_partials(which, gpu_device):
    """BatchNorm partial sums from the LDS-patch stem kernel's epilogue (one row per workgroup; the pixels past a
    ragged last tile are masked): column totals == column sums / sums of squares of the output, output unchanged."""
    from avid_hip import ops
    if which == "audio":
        cin, k, stride, pad, shp = 1, (1, 7, 7), (1, 2, 2), (0, 3, 3), (3, 1, 1, 40, 100)
    else:
        cin, k, stride, pad = 3, (3, 7, 7), (1, 2, 2), (1, 3, 3)
        shp = (2, 3, 4, 24, 28) if which == "video" else (5, 3, 2, 38, 44)     # 19 x 22 = 418 pixels: ragged tiles
    x = T(detgen.det_normalish(f"stembn:{which}:x", shp)).to(gpu_device)
    wd = ops.make_weight(64, cin, *k)
    wd.copy_(T(detgen.det_param(f"stembn:{which}:w.weight", (64, cin) + k)))
    wd = wd.to(gpu_device)
    y = ops.conv_cl(x, wd, stride, pad, channel_first=True)
    y2, part = ops.conv_cl(x, wd, stride, pad, channel_first=True, bn_stats=True)
    assert torch.equal(y2, y)
    assert part.dim() == 3 and part.shape[1:] == (2, 64) and part.shape[0] >= 1
    yd = y2.double().reshape(-1, 64)
    assert relerr(part[:, 0].double().sum(0), yd.sum(0)) < 1e-5 * max(1.0, float(yd.abs().sum(0).max() / (yd.sum(0).abs().max() + 1e-30)))
    assert relerr(part[:, 1].double().sum(0), (yd * yd).sum(0)) < 1e-5


def test_conv_fused_addend(gpu_device):
    from avid_hip import ops
    x = T(detgen.det_normalish("fa:x", (2, 64, 3, 5, 6)))
    r = T(detgen.det_normalish("fa:r", (2, 64, 3, 5, 6)))
    w = T(detgen.det_param("fa:w.weight", (64, 64, 3, 1, 1)))
    wd = ops.make_weight(64, 64, 3, 1, 1)
    wd.copy_(w)
    xd, rd = cl(x).to(gpu_device).requires_grad_(True), cl(r).to(gpu_device).requires_grad_(True)
    y = ops.conv_cl(xd, wd.to(gpu_device), (1, 1, 1), (1, 0, 0), addend=rd)
    yr = F.conv3d(x.double(), w.double(), padding=(1, 0, 0)) + r.double()
    assert relerr(ncdhw(y.detach()), yr) < 2e-5
    g = torch.ones_like(y)
    y.backward(g)
    assert torch.equal(rd.grad, g)


def test_linear_bias_relu(gpu_device):
    from avid_hip import ops
    for (B, cin, cout, relu) in [(5, 512, 128, False), (64, 512, 512, True), (4, 512, 512, True)]:
        x = T(detgen.det_normalish(f"lin:{B}:x", (B, cin)))
        w = T(detgen.det_param(f"lin:{B}:w.weight", (cout, cin)))
        b = T(detgen.det_param(f"lin:{B}:w.bias", (cout,)))
        xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
        yr = F.linear(xr, wr, br)
        yr = F.relu(yr) if relu else yr
        gy = T(detgen.det_uniform(f"lin:{B}:g", (B, cout)))
        (yr * gy.double()).sum().backward()
        xd, wd, bd = (t.to(gpu_device).requires_grad_(True) for t in (x, w, b))
        y = ops.linear(xd, wd, bd, relu)
        y.backward(gy.to(gpu_device))
        assert relerr(y.detach(), yr.detach()) < 2e-5
        assert relerr(xd.grad, xr.grad) < 2e-5
        assert relerr(wd.grad, wr.grad) < 2e-5
        assert relerr(bd.grad, br.grad) < 2e-5


# (up to 8 M elements the finalize runs inside the apply launch, bn_fin_apply / bn_bwd_fin_apply; 140000 x 64 takes the
# separate finalize + apply kernels of the large layers)
@pytest.mark.parametrize("M,C", [(1, 64), (2 * 8 * 28 * 28, 64), (777, 128), (3 * 49 * 2, 256), (40, 512), (100003, 64),
                                 (140000, 64)])
@pytest.mark.parametrize("relu", [False, True])
def test_batchnorm_train(M, C, relu, gpu_device):
    """Train-mode BN(+ReLU) fwd/bwd + running-stat update vs float64 F.batch_norm.  Tol 1e-5 rel."""
    from avid_hip import ops
    if M == 1:
        pytest.skip("torch refuses a single value per channel in train mode")
    x = T(detgen.det_normalish(f"bn:{M}:{C}:x", (M, C))) * 1.7 + 0.3
    g = T(detgen.det_param(f"bn:{M}:{C}:bn.weight", (C,)))
    b = T(detgen.det_param(f"bn:{M}:{C}:bn.bias", (C,)))
    rm = T(detgen.det_param(f"bn:{M}:{C}:bn.running_mean", (C,)))
    rv = T(detgen.det_param(f"bn:{M}:{C}:bn.running_var", (C,)))
    gy = T(detgen.det_uniform(f"bn:{M}:{C}:gy", (M, C)))
    xd, gd, bd = (t.to(gpu_device).requires_grad_(True) for t in (x, g, b))
    rmd, rvd = rm.to(gpu_device), rv.to(gpu_device)
    y = ops.batch_norm_cl(xd.view(1, 1, 1, M, C), gd, bd, rmd, rvd, True, 0.1, 1e-5, relu)
    y.backward(gy.to(gpu_device).view(1, 1, 1, M, C))

    xr, gr, br = (t.double().requires_grad_(True) for t in (x, g, b))
    rmr, rvr = rm.double().clone(), rv.double().clone()
    yr = F.batch_norm(xr.t().unsqueeze(0), rmr, rvr, gr, br, True, 0.1, 1e-5).squeeze(0).t()
    if relu:
        # the sign of a pre-activation within fp32 noise of 0 is implementation-defined: take the
        # device's ReLU pattern, after checking it only disagrees with float64 on such near-ties
        mask = (y.detach().view(M, C) > 0).cpu()
        flips = mask != (yr.detach() > 0)
        assert int(flips.sum()) <= 3 and (not flips.any() or float(yr.detach().abs()[flips].max()) < 1e-5)
        yr = yr * mask.double()
    (yr * gy.double()).sum().backward()
    assert relerr(y.detach().view(M, C), yr.detach()) < 1e-5
    assert relerr(rmd, rmr) < 1e-6 and relerr(rvd, rvr) < 1e-5
    assert relerr(xd.grad, xr.grad) < 2e-5
    assert relerr(gd.grad, gr.grad) < 2e-5 and relerr(bd.grad, br.grad) < 2e-5


def test_batchnorm_eval(gpu_device):
    from avid_hip import ops
    M, C = 500, 128
    x = T(detgen.det_normalish("bne:x", (M, C)))
    g, b = T(detgen.det_param("bne:bn.weight", (C,))), T(detgen.det_param("bne:bn.bias", (C,)))
    rm, rv = T(detgen.det_param("bne:bn.running_mean", (C,))), T(detgen.det_param("bne:bn.running_var", (C,)))
    yr = F.relu(F.batch_norm(x.double(), rm.double(), rv.double(), g.double(), b.double(), False, 0.1, 1e-5))
    with torch.no_grad():
        y = ops.batch_norm_cl(x.to(gpu_device).view(1, 1, 1, M, C), g.to(gpu_device), b.to(gpu_device),
                              rm.to(gpu_device), rv.to(gpu_device), False, 0.1, 1e-5, True)
    assert relerr(y.view(M, C), yr) < 1e-6


def test_maxpool_hw3s2(gpu_device):
    """Bit-exact values; gradient routing identical to ATen CPU incl. ties (post-ReLU zeros)."""
    from avid_hip import ops
    for shp in [(2, 64, 3, 56, 56), (1, 64, 2, 7, 9), (2, 64, 1, 8, 8)]:
        x = F.relu(T(detgen.det_normalish(f"mp:{shp}:x", shp)))          # many exact ties at 0
        xr = x.clone().requires_grad_(True)
        yr = F.max_pool3d(xr, (1, 3, 3), (1, 2, 2), (0, 1, 1))
        gy = T(detgen.det_uniform(f"mp:{shp}:g", tuple(yr.shape)))
        (yr * gy).sum().backward()
        xd = cl(x).to(gpu_device).requires_grad_(True)
        y = ops.maxpool_hw3s2(xd)
        y.backward(cl(gy).to(gpu_device))
        assert torch.equal(ncdhw(y.detach()).cpu(), yr.detach())
        np.testing.assert_allclose(ncdhw(xd.grad).cpu().numpy(), xr.grad.numpy(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("shape", [(2, 3, 10, 12, 64), (1, 2, 7, 9, 64), (3, 2, 56, 56, 64)])
def test_bn_relu_maxpool_fused(shape, gpu_device):
    """The fused stem tail against the three separate ops (already checked against torch): the same pooled
    output, argmax behaviour and running statistics; gradients within 1e-5 (different partial-sum order)."""
    from avid_hip import ops
    B, T_, H, W, C = shape
    x = T(detgen.det_normalish(f"bnpool:{shape}:x", shape)).to(gpu_device)
    g = T(detgen.det_uniform(f"bnpool:{shape}:g", (C,))).to(gpu_device) + 1.5
    b = T(detgen.det_uniform(f"bnpool:{shape}:b", (C,))).to(gpu_device)
    outs = []
    for fused in (False, True):
        xx, gg, bb = x.clone().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
        rm, rv = torch.zeros(C, device=gpu_device), torch.ones(C, device=gpu_device)
        cnt = torch.zeros((), dtype=torch.int64, device=gpu_device)
        if fused:
            y = ops.bn_relu_maxpool(xx, gg, bb, rm, rv, 0.1, 1e-5, cnt)
        else:
            y = ops.maxpool_hw3s2(ops.batch_norm_cl(xx, gg, bb, rm, rv, True, 0.1, 1e-5, True, cnt))
        gy = T(detgen.det_uniform(f"bnpool:{shape}:gy", tuple(y.shape))).to(gpu_device)
        (y * gy).sum().backward()
        outs.append((y.detach(), rm, rv, int(cnt), xx.grad, gg.grad, bb.grad))
    (y0, rm0, rv0, c0, dx0, dg0, db0), (y1, rm1, rv1, c1, dx1, dg1, db1) = outs
    # (small layers fold their partial sums inside the apply launch, in another order than bn_finalize_kernel: the
    # statistics agree to the last bit or two, the pooled output — a max over 9 activations — is compared exactly
    # only where the two normalisations round alike)
    assert relerr(y1, y0) < 1e-6 and relerr(rm1, rm0) < 1e-6 and relerr(rv1, rv0) < 1e-6 and c0 == c1 == 1
    assert relerr(dx1, dx0) < 1e-5 and relerr(dg1, dg0) < 1e-5 and relerr(db1, db0) < 1e-5


def test_global_maxpool(gpu_device):
    from avid_hip import ops
    for shp in [(3, 512, 1, 4, 4), (2, 512, 1, 3, 7), (2, 64, 2, 5, 5)]:
        x = F.relu(T(detgen.det_normalish(f"gp:{shp}:x", shp)))
        xr = x.clone().requires_grad_(True)
        yr = F.adaptive_max_pool3d(xr, (1, 1, 1))
        gy = T(detgen.det_uniform(f"gp:{shp}:g", tuple(yr.shape)))
        (yr * gy).sum().backward()
        xd = cl(x).to(gpu_device).requires_grad_(True)
        y = ops.global_maxpool(xd)
        y.backward(gy.view(shp[0], shp[1]).to(gpu_device))
        assert torch.equal(y.detach().cpu(), yr.detach().view(shp[0], shp[1]))
        assert torch.equal(ncdhw(xd.grad).cpu(), xr.grad)


# ------------------------------------------------------------------------------- criterion ops
def test_l2norm(gpu_device):
    from avid_hip import ops
    x = T(detgen.det_normalish("l2:x", (7, 128))) * 3
    x[3] = 0                                                         # eps clamp row
    xr = x.double().requires_grad_(True)
    yr = F.normalize(xr, p=2, dim=1)
    g = T(detgen.det_uniform("l2:g", (7, 128)))
    (yr * g.double()).sum().backward()
    xd = x.to(gpu_device).requires_grad_(True)
    y = ops.l2_normalize(xd)
    y.backward(g.to(gpu_device))
    assert relerr(y.detach(), yr.detach()) < 1e-6
    mask = torch.arange(7) != 3
    assert relerr(xd.grad[mask], xr.grad[mask]) < 1e-5


def test_alias_draw_bit_exact(gpu_device):
    """Integer path: the HIP draw equals the oracle's Philox restatement bit for bit."""
    from avid_hip import ops
    for probs, n, seed, off in [(None, 100000, 1234, 0), ([.5, .3, .1, .1], 50000, 99, 7),
                                (np.abs(detgen.det_uniform("alias:det50", (50,))) + 0.01, 30000, 2 ** 40 + 5, 2 ** 33)]:
        if probs is None:
            prob, alias = O.alias_build_uniform(1999999)
        else:
            prob, alias = O.alias_build(probs)
        want = O.alias_draw_philox(prob, alias, n, seed, off)
        got = ops.alias_draw(n, len(prob), T(prob).to(gpu_device), T(alias).to(gpu_device), probs is None, seed, off,
                             device=gpu_device)
        np.testing.assert_array_equal(got.cpu().numpy(), want)
    # fused "avoid self" (criterions/avid.py:85) at the AudioSet-scale bank, K = 1024
    N, bs, K = 2_000_000, 64, 1024
    prob, alias = O.alias_build_uniform(N - 1)
    y = T(detgen.det_indices("alias:y", bs, N))
    got = ops.alias_draw(bs * K, N - 1, T(prob).to(gpu_device), T(alias).to(gpu_device), True, 5, 11,
                         y=y.to(gpu_device), per_row=K).view(bs, K).cpu().numpy()
    want = O.sample_negatives_from_draw(O.alias_draw_philox(prob, alias, bs * K, 5, 11), y.numpy(), K)
    np.testing.assert_array_equal(got, want)
    assert (got != y.numpy()[:, None]).all() and got.min() >= 0 and got.max() < N


def test_bank_scores_and_backward(gpu_device):
    from avid_hip import ops
    N, bs, R = 5000, 6, 1025
    bank = F.normalize(T(detgen.det_normalish("bs:bank", (N, 128))), dim=1)
    emb = F.normalize(T(detgen.det_normalish("bs:emb", (bs, 128))), dim=1)
    idx = T(detgen.det_indices("bs:idx", bs * R, N)).view(bs, R)
    er = emb.double().requires_grad_(True)
    sr = torch.bmm(bank.double()[idx], er.unsqueeze(2)).squeeze(-1) / 0.07
    g = T(detgen.det_uniform("bs:g", (bs, R)))
    (sr * g.double()).sum().backward()
    ed = emb.to(gpu_device).requires_grad_(True)
    bank_d = bank.to(gpu_device)
    s = ops.bank_scores(ed, bank_d, idx.to(gpu_device), 1 / 0.07)
    bank_d.mul_(0.0)            # backward must use the PRE-update snapshot, not the live bank
    s.backward(g.to(gpu_device))
    assert relerr(s.detach(), sr.detach()) < 1e-6
    assert relerr(ed.grad, er.grad) < 1e-5


def test_nce_matches_golden(golden, gpu_device):
    """criterions.nce.NCECriterion on the GPU vs the reference's own outputs (tests/golden/nce.npz)."""
    from criterions.nce import NCECriterion
    g = golden("nce")
    for tag, (bs, Pn, K) in {"p1k64": (4, 1, 64), "p32k64": (3, 32, 64)}.items():
        sp = T(detgen.det_uniform(f"nce:{tag}:pos", (bs, Pn)) * 8.0).to(gpu_device).requires_grad_(True)
        sn = T(detgen.det_uniform(f"nce:{tag}:neg", (bs, K)) * 8.0).to(gpu_device).requires_grad_(True)
        crit = NCECriterion(1000).to(gpu_device)
        loss = crit(sp, sn)
        loss.backward()
        np.testing.assert_allclose(loss.item(), g[f"{tag}_loss1"], rtol=2e-6)
        np.testing.assert_allclose(float(crit.avg_exp_score), g[f"{tag}_Z"], rtol=2e-6)
        np.testing.assert_allclose(sp.grad.cpu().numpy(), g[f"{tag}_gpos1"], rtol=2e-5, atol=1e-8)
        np.testing.assert_allclose(sn.grad.cpu().numpy(), g[f"{tag}_gneg1"], rtol=2e-5, atol=1e-8)
        sn.grad = None
        sp2 = (sp.detach() * 0.5).requires_grad_(True)
        loss2 = crit(sp2, sn)
        loss2.backward()
        np.testing.assert_allclose(loss2.item(), g[f"{tag}_loss2"], rtol=2e-6)
        np.testing.assert_allclose(sp2.grad.cpu().numpy(), g[f"{tag}_gpos2"], rtol=2e-5, atol=1e-8)
        np.testing.assert_allclose(sn.grad.cpu().numpy(), g[f"{tag}_gneg2"], rtol=2e-5, atol=1e-8)
        # strided (column-sliced) inputs take the no-copy ld path
        full = torch.cat([sp.detach(), sn.detach()], 1)
        loss3 = crit(full[:, :Pn], full[:, Pn:])
        np.testing.assert_allclose(loss3.item(), float(crit(sp.detach(), sn.detach())), rtol=1e-7)


def test_nce_multi_block_path(gpu_device):
    """At the benchmark size (64 x 1024 negatives) avid_nce_fwd runs 16 blocks whose fp64 partials are summed by
    the last block to finish (device-side ticket, re-armed by the kernel).  vs the oracle's formula in float64;
    repeated calls (ticket re-arm) are bit-identical; P = 32 positives and a ragged K take the same path."""
    from avid_hip import ops
    for bs, Pn, K in ((64, 1, 1024), (64, 32, 1000), (48, 32, 1024)):
        gen = torch.Generator().manual_seed(bs + Pn + K)
        sp = (torch.rand(bs, Pn, generator=gen) * 16 - 8)
        sn = (torch.rand(bs, K, generator=gen) * 16 - 8)
        Z = torch.tensor(0.37)
        want, _ = O.nce_loss(sp.double(), sn.double(), Z.double())
        spd, snd, Zd = sp.to(gpu_device).requires_grad_(True), sn.to(gpu_device).requires_grad_(True), Z.to(gpu_device)
        losses = [ops.nce_loss(spd, snd, Zd) for _ in range(5)]
        np.testing.assert_allclose(losses[0].item(), float(want), rtol=2e-6)
        assert all(torch.equal(l, losses[0]) for l in losses[1:])
        losses[-1].backward()
        spr, snr = sp.double().requires_grad_(True), sn.double().requires_grad_(True)
        O.nce_loss(spr, snr, Z.double())[0].backward()
        np.testing.assert_allclose(spd.grad.cpu().numpy(), spr.grad.float().numpy(), rtol=2e-5, atol=1e-9)
        np.testing.assert_allclose(snd.grad.cpu().numpy(), snr.grad.float().numpy(), rtol=2e-5, atol=1e-9)


def test_nce_joint_score_tensor_path(gpu_device):
    """ops.split_scores tags the [positives | negatives] views of one bank_scores result; nce_loss then differentiates
    the joint tensor (one gradient tensor, no zero-fill / slice-copy / add kernels from autograd).  Loss and the
    gradient of the joint tensor are bit-identical to the path through two plain views."""
    from avid_hip import ops
    for bs, Pn, K in ((64, 1, 1024), (16, 32, 64)):
        gen = torch.Generator().manual_seed(7 * bs + Pn)
        base = (torch.rand(bs, Pn + K, generator=gen) * 12 - 6).to(gpu_device)
        Z = torch.tensor(0.41, device=gpu_device)
        s1 = base.clone().requires_grad_(True)
        l1 = ops.nce_loss(s1[:, :Pn], s1[:, Pn:], Z)                 # plain views: the separate-tensor path
        (l1 * 0.5).backward()
        s2 = base.clone().requires_grad_(True)
        t = s2 * 1.0                                                # a non-leaf, like a bank_scores result
        pos, neg = ops.split_scores(t, Pn)
        assert pos._avid_joint is t and neg._avid_joint is t
        l2 = ops.nce_loss(pos, neg, Z)
        assert type(l2.grad_fn).__name__.startswith("_NCELossJoint")
        (l2 * 0.5).backward()
        assert torch.equal(l1, l2) and torch.equal(s1.grad, s2.grad)


def test_bank_update(gpu_device):
    from avid_hip import ops
    N, B = 3000, 40
    bank = F.normalize(T(detgen.det_normalish("bu:bank", (N, 128))), dim=1)
    emb = F.normalize(T(detgen.det_normalish("bu:emb", (B, 128))), dim=1)
    y = T(detgen.det_indices("bu:y", B, N))
    y[7] = y[3]
    y[30] = y[3]                                    # duplicates: last occurrence (30) wins
    v1, v2 = bank.clone(), bank.clone()
    O.update_memory(v1, v2, emb, emb, y, (0.5, 0.9))
    b1, b2 = bank.to(gpu_device), bank.to(gpu_device)
    ops.bank_update(b1, y.to(gpu_device), emb.to(gpu_device), 0.5)
    ops.bank_update(b2, y.to(gpu_device), emb.to(gpu_device), 0.9)
    np.testing.assert_allclose(b1.cpu().numpy(), v1.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(b2.cpu().numpy(), v2.numpy(), rtol=1e-6, atol=1e-7)
    untouched = np.setdiff1d(np.arange(N), y.numpy())
    assert torch.equal(b1.cpu()[untouched], bank[untouched])        # bit-exact elsewhere
    np.testing.assert_allclose(b1.cpu()[y].norm(dim=1).numpy(), 1.0, rtol=1e-6)


@pytest.mark.parametrize("bs,K,N", [(6, 1024, 5000), (64, 1024, 240000), (3, 70, 300)])
def test_xmodal_fused_vs_unfused_ops_and_fp64(bs, K, N, gpu_device):
    """ops.xmodal_fused (normalize -> gather both banks -> scores / T -> NCE -> gradient, one kernel) against (a) the
    chain of unfused ops it replaces and (b) the same arithmetic in float64 on the CPU (criterions/avid.py:52-71,
    criterions/nce.py:38-58 with a frozen Z): losses 2e-6, gradients 2e-5 of their scale; normalised embeddings 1e-6;
    repeated calls bit-identical (fixed summation order, re-armed tickets); bank_update_pair == two bank_update."""
    from avid_hip import ops
    gen = torch.Generator().manual_seed(bs * 1000 + K)
    v1 = F.normalize(torch.randn(N, 128, generator=gen), dim=1)
    v2 = F.normalize(torch.randn(N, 128, generator=gen), dim=1)
    ve = torch.randn(bs, 128, generator=gen) * 2
    ae = torch.randn(bs, 128, generator=gen) * 0.5
    y = torch.randperm(N, generator=gen)[:bs]
    idx = torch.randint(0, N - 1, (bs, K), generator=gen)
    idx = idx + (idx >= y[:, None]).long()
    Z, coeff, T_ = torch.tensor(0.83), 0.75, 0.07
    # (b) float64
    vr, ar = ve.double().requires_grad_(True), ae.double().requires_grad_(True)
    vh, ah = F.normalize(vr, dim=1), F.normalize(ar, dim=1)
    rows = torch.cat([y[:, None], idx], 1)
    s_v2a = torch.bmm(v2.double()[rows], vh.unsqueeze(2)).squeeze(-1) / T_
    s_a2v = torch.bmm(v1.double()[rows], ah.unsqueeze(2)).squeeze(-1) / T_
    l1, _ = O.nce_loss(s_v2a[:, :1], s_v2a[:, 1:], Z.double())
    l2, _ = O.nce_loss(s_a2v[:, :1], s_a2v[:, 1:], Z.double())
    tot = (l1 / 2 + l2 / 2) * coeff
    tot.backward()
    # (a) device
    b1, b2 = v1.to(gpu_device), v2.to(gpu_device)
    vd, ad = ve.to(gpu_device).requires_grad_(True), ae.to(gpu_device).requires_grad_(True)
    ws = ops.xmodal_fused_workspace(gpu_device, bs, K)
    outs = [ops.xmodal_fused(vd, ad, y.to(gpu_device), idx.to(gpu_device), b1, b2, Z.to(gpu_device), 1 / T_, coeff, ws)
            for _ in range(3)]
    total, losses, hats = outs[0]
    assert all(torch.equal(o[0], total) and torch.equal(o[1], losses) and torch.equal(o[2], hats) for o in outs[1:])
    (total * 2.0).backward()                                         # the upstream gradient scales the stored one
    np.testing.assert_allclose(losses.cpu().numpy(), [float(l1), float(l2), float(l1 / 2 + l2 / 2), float(tot)], rtol=2e-6)
    assert relerr(hats[0], vh.detach()) < 1e-6 and relerr(hats[1], ah.detach()) < 1e-6
    assert relerr(vd.grad, 2.0 * vr.grad) < 2e-5 and relerr(ad.grad, 2.0 * ar.grad) < 2e-5
    # unfused chain on the device
    vu, au = ve.to(gpu_device).requires_grad_(True), ae.to(gpu_device).requires_grad_(True)
    vhu, ahu = ops.l2_normalize(vu), ops.l2_normalize(au)
    rows_d = rows.to(gpu_device)
    su1 = ops.bank_scores(vhu, b2, rows_d, 1 / T_)
    su2 = ops.bank_scores(ahu, b1, rows_d, 1 / T_)
    lu = (ops.nce_loss(*ops.split_scores(su1, 1), Z.to(gpu_device)) / 2 + ops.nce_loss(*ops.split_scores(su2, 1), Z.to(gpu_device)) / 2) * coeff
    (lu * 2.0).backward()
    np.testing.assert_allclose(float(total), float(lu), rtol=2e-6)
    assert relerr(vd.grad, vu.grad) < 2e-5 and relerr(ad.grad, au.grad) < 2e-5
    # both banks in one launch == two launches, bit for bit (incl. a duplicate id)
    y2 = y.clone()
    if bs > 2:
        y2[2] = y2[0]
    p1, p2, q1, q2 = b1.clone(), b2.clone(), b1.clone(), b2.clone()
    ops.bank_update_pair(p1, p2, y2.to(gpu_device), hats[0], hats[1], 0.5, 0.9)
    ops.bank_update(q1, y2.to(gpu_device), hats[0], 0.5)
    ops.bank_update(q2, y2.to(gpu_device), hats[1], 0.9)
    assert torch.equal(p1, q1) and torch.equal(p2, q2)
    ops.check_device_errors(gpu_device)


@pytest.mark.parametrize("bs,P,K,Kw,N", [(6, 32, 1024, 64, 5000), (64, 32, 1024, 64, 240000), (3, 5, 70, 70, 300), (4, 1, 33, 7, 200)])
def test_cma_fused_vs_fp64_and_the_criterion_it_replaces(bs, P, K, Kw, N, gpu_device, monkeypatch):
    """ops.cma_fused (the AVID+CMA criterion's stock term set in one kernel: criterions/avid_cma.py:150-194, 338-358 +
    criterions/nce.py:38-58 with a frozen Z) against the same arithmetic in float64 on the CPU — four losses, the two
    group means and the total 2e-6, gradients 2e-5 of their scale, normalised embeddings 1e-6, repeated calls
    bit-identical — and, through criterions.AVID_CMA, against the chain of unfused ops (AVID_FUSED_CRITERION=0)."""
    from avid_hip import ops
    gen = torch.Generator().manual_seed(bs * 1000 + K + P)
    v1 = F.normalize(torch.randn(N, 128, generator=gen), dim=1)
    v2 = F.normalize(torch.randn(N, 128, generator=gen), dim=1)
    ve = torch.randn(bs, 128, generator=gen) * 2
    ae = torch.randn(bs, 128, generator=gen) * 0.5
    y = torch.randperm(N, generator=gen)[:bs]
    pos = torch.randint(0, N, (bs, P), generator=gen)
    idx = torch.randint(0, N, (bs, K), generator=gen)
    Z, cI, cP, T_ = torch.tensor(0.83), 0.4, 0.6, 0.07
    vr, ar = ve.double().requires_grad_(True), ae.double().requires_grad_(True)
    vh, ah = F.normalize(vr, dim=1), F.normalize(ar, dim=1)
    rows = torch.cat([y[:, None], pos, idx], 1)
    sc = lambda bank, e: torch.bmm(bank.double()[rows], e.unsqueeze(2)).squeeze(-1) / T_
    s_v2a, s_a2v, s_v2v, s_a2a = sc(v2, vh), sc(v1, ah), sc(v1, vh), sc(v2, ah)
    neg = slice(1 + P, None)
    wneg = slice(1 + P, 1 + P + Kw)
    l = [O.nce_loss(s_v2a[:, :1], s_v2a[:, neg], Z.double())[0], O.nce_loss(s_a2v[:, :1], s_a2v[:, neg], Z.double())[0],
         O.nce_loss(s_v2v[:, 1:1 + P], s_v2v[:, wneg], Z.double())[0], O.nce_loss(s_a2a[:, 1:1 + P], s_a2a[:, wneg], Z.double())[0]]
    gi, gp = l[0] / 2 + l[1] / 2, l[2] / 2 + l[3] / 2
    tot = gi * cI + gp * cP
    tot.backward()
    b1, b2 = v1.to(gpu_device), v2.to(gpu_device)
    vd, ad = ve.to(gpu_device).requires_grad_(True), ae.to(gpu_device).requires_grad_(True)
    ws = ops.cma_fused_workspace(gpu_device, bs, P, K)
    outs = [ops.cma_fused(vd, ad, y.to(gpu_device), pos.to(gpu_device), idx.to(gpu_device), b1, b2, Z.to(gpu_device), 1 / T_,
                          Kw, cI, cP, ws) for _ in range(3)]
    total, losses, hats = outs[0]
    assert all(torch.equal(o[0], total) and torch.equal(o[1][:7], losses[:7]) and torch.equal(o[2], hats) for o in outs[1:])
    (total * 2.0).backward()
    np.testing.assert_allclose(losses[:7].cpu().numpy(), [float(v) for v in l] + [float(gi), float(gp), float(tot)], rtol=2e-6)
    assert relerr(hats[0], vh.detach()) < 1e-6 and relerr(hats[1], ah.detach()) < 1e-6
    assert relerr(vd.grad, 2.0 * vr.grad) < 2e-5 and relerr(ad.grad, 2.0 * ar.grad) < 2e-5
    ops.check_device_errors(gpu_device)


def test_avid_cma_criterion_fused_step_equals_the_unfused_one(gpu_device, monkeypatch):
    """criterions.AVID_CMA: the second step (Z frozen) through ops.cma_fused against the same step with
    AVID_FUSED_CRITERION=0 — same negatives (same sampler stream), loss and tb_log 3e-6, embedding gradients 2e-5, banks
    after the update 1e-6."""
    import criterions
    from avid_hip import ops
    res = []
    for fused in (True, False):
        monkeypatch.setattr(ops, "FUSED_CRITERION", fused)
        torch.manual_seed(5)
        c = criterions.AVID_CMA(num_data=3000, embedding_dim=128, num_negatives=256, num_negatives_within=32, momentum=0.5,
                                xModalInstCoeff=1., wModalInstCoeff=0., xModalPosCoeff=0., wModalPosCoeff=1.,
                                sampling_args={"type": "consensus", "pos_k": 8}, device=gpu_device.index or 0)
        g = torch.Generator().manual_seed(11)
        out = []
        for step in range(2):
            v = torch.randn(6, 128, generator=g).to(gpu_device).requires_grad_(True)
            a = torch.randn(6, 128, generator=g).to(gpu_device).requires_grad_(True)
            yy = torch.randperm(3000, generator=g)[:6].to(gpu_device)
            loss, tb = c(v, a, yy)
            loss.backward()
            out.append((float(loss), {k: float(x) for k, x in tb.items()}, v.grad.clone(), a.grad.clone()))
        res.append((out, c.nce_average.view1_mem.clone(), c.nce_average.view2_mem.clone()))
    (f, fb1, fb2), (u, ub1, ub2) = res
    assert f[0][0] == u[0][0]                              # the first step is the unfused path either way
    np.testing.assert_allclose(f[1][0], u[1][0], rtol=3e-6)
    assert set(f[1][1]) == set(u[1][1])
    for k in f[1][1]:
        np.testing.assert_allclose(f[1][1][k], u[1][1][k], rtol=3e-6)
    assert relerr(f[1][2], u[1][2]) < 2e-5 and relerr(f[1][3], u[1][3]) < 2e-5
    assert relerr(fb1, ub1) < 1e-6 and relerr(fb2, ub2) < 1e-6


def test_cma_negatives_bit_exact(golden, gpu_device):
    from avid_hip import ops
    g = golden("cma")
    pset = T(g["topk_consensus"]).int().to(gpu_device)
    pos, neg = ops.cma_negatives(pset, T(g["ms_y"]).to(gpu_device), T(g["ms_rand"]).to(gpu_device))
    np.testing.assert_array_equal(pos.cpu().numpy(), g["ms_pos"])
    np.testing.assert_array_equal(neg.cpu().numpy(), g["ms_neg"])


def test_adam_flat(gpu_device):
    from avid_hip import ops
    n = 100003
    p = T(detgen.det_normalish("adam:p", (n,)))
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=2e-4, weight_decay=1e-5)
    pd = p.to(gpu_device)
    m, v = torch.zeros_like(pd), torch.zeros_like(pd)
    pd2, m2, v2 = pd.clone(), torch.zeros_like(pd), torch.zeros_like(pd)      # device-resident step counter path
    t_dev = torch.zeros((), dtype=torch.int64, device=gpu_device)
    for step in range(1, 4):
        g = T(detgen.det_normalish(f"adam:g{step}", (n,)))
        ref.grad = g.clone()
        opt.step()
        ops.adam_flat(pd, g.to(gpu_device), m, v, 2e-4, 0.9, 0.999, 1e-8, 1e-5, step)
        ops.adam_flat(pd2, g.to(gpu_device), m2, v2, 2e-4, 0.9, 0.999, 1e-8, 1e-5, 0, step_dev=t_dev)
    np.testing.assert_allclose(pd.cpu().numpy(), ref.detach().numpy(), rtol=1e-6, atol=1e-7)
    assert int(t_dev) == 3
    np.testing.assert_allclose(pd2.cpu().numpy(), pd.cpu().numpy(), rtol=1e-6, atol=1e-7)


def test_winograd_random_shapes(gpu_device):
    """24 random layers (64-256 channels incl. the 3-column-block case 192, 2-40 pixel extents, 1-5 x 1-4 frames,
    addend / statistics epilogues at random) through wino_kernel or wino2_kernel, the input gradient through the same
    kernel and the weight gradient through wino_wgrad_kernel, against float64 F.conv3d: 2e-5 of the scale
    (tools/wino_fuzz.py is the long version: 80 shapes, worst 8e-7)."""
    import random
    from avid_hip import ops
    rng = random.Random(11)
    ops.wino_configure(1, 1, 256)
    try:
        for it in range(24):
            cin, cout = rng.choice([64, 128, 192, 256]), rng.choice([64, 128, 192, 256])
            B, T_, H, W = rng.randint(1, 5), rng.randint(1, 4), rng.randint(2, 40), rng.randint(2, 40)
            ops.wino2_configure(0 if rng.random() < 0.6 else 100000)
            g = torch.Generator().manual_seed(it)
            x = torch.randn(B, T_, H, W, cin, generator=g).to(gpu_device).requires_grad_(True)
            w = ops.make_weight(cout, cin, 1, 3, 3)
            w.copy_(torch.randn(cout, cin, 1, 3, 3, generator=g))
            w = w.to(gpu_device).requires_grad_(True)
            add = torch.randn(B, T_, H, W, cout, generator=g).to(gpu_device) if rng.random() < 0.5 else None
            stats = rng.random() < 0.5
            out = ops.conv_cl(x, w, (1, 1, 1), (0, 1, 1), addend=add, bn_stats=stats)
            y, part = out if stats else (out, None)
            gy = torch.randn(y.shape, generator=g).to(gpu_device)
            y.backward(gy)
            xr = x.detach().double().permute(0, 4, 1, 2, 3).requires_grad_(True)
            wr = w.detach().double().requires_grad_(True)
            yr = F.conv3d(xr, wr, padding=(0, 1, 1)).permute(0, 2, 3, 4, 1)
            if add is not None:
                yr = yr + add.double()
            (yr * gy.double()).sum().backward()
            tag = (it, B, T_, H, W, cin, cout)
            assert relerr(y.detach(), yr.detach()) < 2e-5, tag
            assert relerr(x.grad, xr.grad.permute(0, 2, 3, 4, 1)) < 2e-5, tag
            assert relerr(w.grad, wr.grad) < 5e-5, tag
            if stats and part is not None and part.numel():
                yd = y.detach().double().reshape(-1, cout)
                assert relerr(part[:, 1].double().sum(0), (yd * yd).sum(0)) < 1e-5, tag
    finally:
        ops.wino_configure(-1, -1, -1)
        ops.wino2_configure(-1)


def test_winograd_bn_backward_sums_random_shapes(gpu_device):
    """conv1 -> BN+ReLU -> conv2 (3x3, on wino_kernel or wino2_kernel) [+ a second consumer through the tap]: the
    BatchNorm-backward sums from conv2's input-gradient epilogue (wino*_kernel<4> / <6>) against the chain without the
    hand-over, for 16 random layers (64-256 channels, odd / tiny extents): 2e-5 of each gradient's scale."""
    import random
    from avid_hip import ops
    rng = random.Random(23)
    ops.wino_configure(1, 1, 256)
    try:
        for it in range(16):
            cmid, cout = rng.choice([64, 128, 256]), rng.choice([64, 128, 192, 256])      # (BatchNorm: power-of-two channels)
            B, T_, H, W = rng.randint(1, 4), rng.randint(1, 3), rng.randint(3, 30), rng.randint(3, 30)
            ops.wino2_configure(0 if rng.random() < 0.6 else 100000)
            tap = rng.random() < 0.5
            g = torch.Generator().manual_seed(100 + it)
            x = torch.randn(B, T_, H, W, 64, generator=g).to(gpu_device)
            w1 = ops.make_weight(cmid, 64, 1, 3, 3); w1.copy_(torch.randn(cmid, 64, 1, 3, 3, generator=g) * 0.1)
            w2 = ops.make_weight(cout, cmid, 1, 3, 3); w2.copy_(torch.randn(cout, cmid, 1, 3, 3, generator=g) * 0.1)
            w1, w2 = w1.to(gpu_device), w2.to(gpu_device)
            gam = (torch.rand(cmid, generator=g) + 0.5).to(gpu_device)
            bet = (torch.rand(cmid, generator=g) - 0.5).to(gpu_device)
            gy, res = None, []
            for fused in (False, True):
                xx = x.clone().requires_grad_(True)
                g_, b_ = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
                rm, rv = torch.zeros(cmid, device=gpu_device), torch.ones(cmid, device=gpu_device)
                y1 = ops.conv_cl(xx, w1, (1, 1, 1), (0, 1, 1))
                src = ops.BnSource(None, None, True) if fused else None
                hdn = ops.batch_norm_cl(y1, g_, b_, rm, rv, True, relu=True, src=src)
                out = ops.conv_cl(hdn, w2, (1, 1, 1), (0, 1, 1), tap=tap, bn_src=src)
                y2, alias = (out[0], out[-1]) if tap else (out, None)
                if gy is None:
                    gy = torch.randn(y2.shape, generator=g).to(gpu_device)
                loss = (y2 * gy).sum()
                if tap:
                    loss = loss + (alias * alias).sum() * 0.25
                loss.backward()
                res.append((xx.grad.clone(), g_.grad.clone(), b_.grad.clone()))
            for a, b in zip(res[0], res[1]):
                assert relerr(b, a) < 2e-5, (it, B, T_, H, W, cmid, cout, tap)
    finally:
        ops.wino_configure(-1, -1, -1)
        ops.wino2_configure(-1)


@pytest.mark.parametrize("shape,cin,cout", [((6, 8, 27, 29), 64, 64), ((5, 4, 45, 47), 128, 128), ((8, 2, 21, 19), 256, 256),
                                            ((64, 1, 10, 25), 64, 64), ((3, 8, 56, 56), 64, 128)])
def test_wino2_presplit_is_bit_identical(shape, cin, cout, gpu_device, kernel_log):
    """wino2p_kernel (V split once by the transform thread, three bf16 planes in a ring of half-stages; default) against
    wino2_kernel (every product wave splits its fragments): the same split of the same values and the same six products in
    the same order — output, BatchNorm partial sums (with and without the residual addend) and the input gradient with the
    BatchNorm-backward sums out of its epilogue are bit-identical.  Odd extents, one to four column blocks, 4 to 16 chunks."""
    from avid_hip import ops, lib
    B, Ti, Hi, Wi = shape
    x = T(detgen.det_normalish(f"w2p:{shape}:{cin}:x", (B, Ti, Hi, Wi, cin))).to(gpu_device)
    w = ops.make_weight(cout, cin, 1, 3, 3)
    w.copy_(T(detgen.det_param(f"w2p:{cout}:{cin}:w.weight", (cout, cin, 1, 3, 3))))
    w = w.to(gpu_device)
    add = T(detgen.det_uniform(f"w2p:{shape}:{cout}:add", (B, Ti, Hi, Wi, cout))).to(gpu_device)
    gy = T(detgen.det_uniform(f"w2p:{shape}:{cout}:gy", (B, Ti, Hi, Wi, cout))).to(gpu_device)
    gam = (T(detgen.det_uniform(f"w2p:{cin}:g", (cin,))) + 1.5).to(gpu_device)
    bet = T(detgen.det_uniform(f"w2p:{cin}:b", (cin,))).to(gpu_device)
    pre = lib.raw("avid_wino2_pre_configure")
    ops.wino_configure(1, 1, 256)
    ops.wino2_configure(0)
    res = {}
    try:
        for on in (1, 0):
            pre(on)
            with kernel_log() as log:
                out = [ops.conv_cl(x, w, (1, 1, 1), (0, 1, 1))]
                out += list(ops.conv_cl(x, w, (1, 1, 1), (0, 1, 1), bn_stats=True))
                out += list(ops.conv_cl(x, w, (1, 1, 1), (0, 1, 1), addend=add, bn_stats=True))
                # BN + ReLU in front of the conv: its input gradient carries the BatchNorm-backward sums (EPI 4)
                xx = x.clone().requires_grad_(True)
                g_, b_ = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
                rm, rv = torch.zeros(cin, device=gpu_device), torch.ones(cin, device=gpu_device)
                src = ops.BnSource(None, None, True)
                hh = ops.batch_norm_cl(xx, g_, b_, rm, rv, True, relu=True, src=src)
                y2 = ops.conv_cl(hh, w, (1, 1, 1), (0, 1, 1), bn_src=src)
                (y2 * gy).sum().backward()
                out += [y2.detach(), xx.grad, g_.grad, b_.grad]
            # which form ran, from the launch log: wino2p_kernel (V split once) with the switch on, wino2_kernel with it off
            assert log.launches("wino2p_kernel" if on else "wino2_kernel") >= 5 and log.launches("wino2_kernel" if on else "wino2p_kernel") == 0 \
                and log.launches("wino_kernel") == 0, sorted(log.report)
            res[on] = [t.clone() for t in out]
    finally:
        pre(-1)
        ops.wino_configure(-1, -1, -1)
        ops.wino2_configure(-1)
    for a, b in zip(res[1], res[0]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("M,C", [(64 * 8 * 28 * 28, 64), (1024, 512), (777, 128)], ids=["conv2x", "conv5x", "ragged"])
def test_batchnorm_statistics_only(M, C, gpu_device, kernel_log):
    """avid_bn_fwd_train(y = NULL): the saved vectors, running statistics and the batch counter of the full call, and no apply pass —
    what the launch programs emit in front of a convolution that applies the map itself (avid_conv_fwd_in).  Large layers: the
    same finalize launch, bit-identical vectors; small layers (whose full call folds inside its apply launch) agree to rounding."""
    from avid_hip import lib, ops
    x = (T(detgen.det_normalish(f"bnso:{M}:{C}:x", (M, C))) * 1.3 - 0.2).to(gpu_device)
    g = T(detgen.det_param(f"bnso:{M}:{C}:bn.weight", (C,))).to(gpu_device)
    b = T(detgen.det_param(f"bnso:{M}:{C}:bn.bias", (C,))).to(gpu_device)
    outs = []
    for full in (True, False):
        rm, rv = torch.zeros(C, device=gpu_device), torch.ones(C, device=gpu_device)
        nbt = torch.zeros((), dtype=torch.int64, device=gpu_device)
        s4 = torch.empty(4, C, device=gpu_device)
        y = torch.empty_like(x) if full else None
        ws = ops.workspace(gpu_device, ops._bn_ws_bytes(M, C))
        with kernel_log() as log:
            lib.call("avid_bn_fwd_train", M, C, ops._p(x), ops._p(g), ops._p(b), ops._p(rm), ops._p(rv), 0.1, 1e-5, 1, ops._p(y),
                     ops._p(s4[0]), ops._p(s4[1]), ops._p(s4[2]), ops._p(s4[3]), ops._p(nbt), None, 0, ops._p(ws), ws.numel(), ops._stream())
        if not full:
            assert log.launches("bn_apply_kernel") == 0 and log.launches("bn_fin_apply_kernel") == 0, sorted(log.report)
        outs.append((s4, rm, rv, int(nbt)))
    (s4a, rma, rva, na), (s4b, rmb, rvb, nb) = outs
    assert na == nb == 1
    if M * C > (1 << 23):              # the full call ran the same finalize launch
        assert torch.equal(s4a, s4b) and torch.equal(rma, rmb) and torch.equal(rva, rvb)
    else:
        assert relerr(s4b, s4a) < 1e-6 and relerr(rmb, rma) < 1e-6 and relerr(rvb, rva) < 1e-6
