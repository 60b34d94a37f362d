"""tconv64_kernel / twgrad64_kernel — conv2x's (3,1,1) stride-1 layers (models/network_blocks.py:37,42: 64 -> 64 channels, 8 frames) with
every input row staged once for its three taps and the pre-split weights resident in LDS — against float64 `F.conv3d`,
against igemm_pk_kernel on the same layer, and through every epilogue it carries (BatchNorm partial sums, residual
addend, BatchNorm-backward sums).  `avid_tconv_configure(2)` sends the small fixtures through it (the default rule asks for
three rounds of tiles for the CUs)."""
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


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture
def tconv(gpu_device):
    from avid_hip import lib
    conf = lib.raw("avid_tconv_configure")

    def set_mode(m):
        return conf(m)
    assert set_mode(2) == 2
    yield set_mode
    set_mode(-1)


# (B, H, W) at 8 frames: a ragged last tile (585 positions = 18 tiles + 9), tiles that straddle clips (400 = 12.5 tiles
# per clip), whole tiles only, fewer tiles than one
SHAPES = [(3, 13, 15), (9, 20, 20), (2, 28, 28), (1, 3, 5)]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_tconv_fwd_dgrad_vs_float64_and_igemm(shape, gpu_device, kernel_log, tconv):
    from avid_hip import ops
    B, Hi, Wi = shape
    k, stride, pad = (3, 1, 1), (1, 1, 1), (1, 0, 0)
    x = T(detgen.det_normalish(f"tconv:{shape}:x", (B, 64, 8, Hi, Wi)))
    w = T(detgen.det_param(f"tconv:{shape}:w.weight", (64, 64) + k))
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv3d(xr, wr, stride=stride, padding=pad)
    gy = T(detgen.det_uniform(f"tconv:{shape}:gy", tuple(yr.shape)))
    (yr * gy.double()).sum().backward()
    outs = {}
    for mode in (2, 0):
        tconv(mode)
        xd = cl(x).to(gpu_device).requires_grad_(True)
        wd = ops.make_weight(64, 64, *k)
        wd.copy_(w)
        wd = wd.to(gpu_device).requires_grad_(True)
        with kernel_log() as log:
            y = ops.conv_cl(xd, wd, stride, pad)
            y.backward(cl(gy).to(gpu_device))
        n = log.launches("tconv64_kernel<0>"), log.launches("tconv64_kernel<1>"), log.launches("twgrad64_kernel")
        assert n == ((1, 1, 1) if mode == 2 else (0, 0, 0)), (mode, sorted(log.report))
        outs[mode] = (ncdhw(y.detach()).cpu(), ncdhw(xd.grad).cpu(), wd.grad.cpu())
        assert relerr(outs[mode][0], yr.detach()) < 2e-5
        assert relerr(outs[mode][1], xr.grad) < 2e-5
        assert relerr(outs[mode][2], wr.grad) < 5e-5
    # same six-product arithmetic, another summation order over the taps / channel blocks (weight gradient: over the pixels)
    assert relerr(outs[2][0], outs[0][0]) < 2e-6 and relerr(outs[2][1], outs[0][1]) < 2e-6
    assert relerr(outs[2][2], outs[0][2]) < 1e-5


@pytest.mark.parametrize("shape", SHAPES[:3], ids=lambda s: "x".join(map(str, s)))
def test_tconv_bn_partials_and_addend(shape, gpu_device, kernel_log, tconv):
    """Forward epilogues: BatchNorm partial sums (one row per workgroup + zero rows up to avid_conv_fwd_stats_rows) whose
    column totals are the column sums / sums of squares of the output, with and without the residual addend; the output
    is bit-identical with and without the statistics, and the addend is added exactly once."""
    from avid_hip import ops
    B, Hi, Wi = shape
    x = T(detgen.det_normalish(f"tconvbn:{shape}:x", (B, 8, Hi, Wi, 64))).to(gpu_device)
    w = ops.make_weight(64, 64, 3, 1, 1)
    w.copy_(T(detgen.det_param(f"tconvbn:{shape}:w.weight", (64, 64, 3, 1, 1))))
    w = w.to(gpu_device)
    add = T(detgen.det_uniform(f"tconvbn:{shape}:add", (B, 8, Hi, Wi, 64))).to(gpu_device)
    plain = ops.conv_cl(x, w, (1, 1, 1), (1, 0, 0))
    for addend in (None, add):
        with kernel_log() as log:
            y, part = ops.conv_cl(x, w, (1, 1, 1), (1, 0, 0), addend=addend, bn_stats=True)
        assert log.launches("tconv64_kernel<0>") == 1
        y_plain = ops.conv_cl(x, w, (1, 1, 1), (1, 0, 0), addend=addend)
        assert torch.equal(y, y_plain)
        if addend is not None:
            assert relerr(y, plain + addend) < 1e-6
        assert part.dim() == 3 and part.shape[1:] == (2, 64)
        yd = y.double().reshape(-1, 64)
        s, q = part[:, 0].double().sum(0), part[:, 1].double().sum(0)
        assert relerr(s, yd.sum(0)) < 1e-5 * max(1.0, float(yd.abs().sum(0).max() / (yd.sum(0).abs().max() + 1e-30)))
        assert relerr(q, (yd * yd).sum(0)) < 1e-5


@pytest.mark.parametrize("shape", [(3, 13, 15), (2, 28, 28)], ids=lambda s: "x".join(map(str, s)))
def test_tconv_bn_backward_partials(shape, gpu_device, kernel_log, tconv):
    """conv1 -> BN+ReLU -> temporal conv2 [+ tap]: the BatchNorm's backward partial sums out of tconv64_kernel<1>'s
    epilogue (ops.BnSource) against the BatchNorm's own pass over dy and x — 2e-5 of each gradient's scale; with the tap
    (a second consumer) the hand-over is refused and the result is unchanged."""
    from avid_hip import ops
    B, Hi, Wi = shape
    x = T(detgen.det_normalish(f"tconvbnb:{shape}:x", (B, 8, Hi, Wi, 64))).to(gpu_device)
    w1 = ops.make_weight(64, 64, 1, 3, 3); w1.copy_(T(detgen.det_param("tconvbnb:w1.weight", (64, 64, 1, 3, 3))))
    w2 = ops.make_weight(64, 64, 3, 1, 1); w2.copy_(T(detgen.det_param("tconvbnb:w2.weight", (64, 64, 3, 1, 1))))
    w1, w2 = w1.to(gpu_device), w2.to(gpu_device)
    gam = (T(detgen.det_uniform("tconvbnb:g", (64,))) + 1.5).to(gpu_device)
    bet = T(detgen.det_uniform("tconvbnb:b", (64,))).to(gpu_device)
    gy = None
    res = {}
    for fused in (False, True):
        for tap in (False, True):
            xx = x.clone().requires_grad_(True)
            g_, b_ = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
            rm, rv = torch.zeros(64, device=gpu_device), torch.ones(64, device=gpu_device)
            y1 = ops.conv_cl(xx, w1, (1, 1, 1), (0, 1, 1))
            src = ops.BnSource(None, None, True) if fused else None
            hh = ops.batch_norm_cl(y1, g_, b_, rm, rv, True, relu=True, src=src)
            out = ops.conv_cl(hh, w2, (1, 1, 1), (1, 0, 0), tap=tap, bn_src=src)
            y2, alias = (out[0], out[-1]) if tap else (out, None)
            if gy is None:
                gy = T(detgen.det_uniform(f"tconvbnb:{shape}:gy", tuple(y2.shape))).to(gpu_device)
            loss = (y2 * gy).sum()
            if tap:
                loss = loss + (alias * alias).sum() * 0.25
            with kernel_log() as log:
                loss.backward()
            assert log.launches("tconv64_kernel<1>") == 1, sorted(log.report)
            if fused and not tap:
                assert log.launches("bn_bwd_partial_kernel") == 0       # the sums came from the dgrad's epilogue
            res[(fused, tap)] = (xx.grad.clone(), g_.grad.clone(), b_.grad.clone())
    for tap in (False, True):
        for a, b in zip(res[(False, tap)], res[(True, tap)]):
            assert relerr(b, a) < 2e-5


def test_tconv_dispatch_rule(gpu_device, kernel_log, tconv):
    """Default rule (mode 1): layers with at least three rounds of 32-position tiles for the CUs take it — conv2x at the
    benchmark batch does, the two-clip fixtures do not; other geometries (7 frames, 128 channels, stride 2) never do."""
    from avid_hip import ops
    tconv(1)
    cus = ops.cu_budget()
    big_b = (3 * cus * 32 + 783) // 784           # clips of 28 x 28 for three rounds
    for (B, T_, Hi, Wi, c, stride, want) in [(big_b, 8, 28, 28, 64, (1, 1, 1), 1), (2, 8, 28, 28, 64, (1, 1, 1), 0)]:
        x = torch.randn(B, T_, Hi, Wi, c, device=gpu_device)
        w = ops.make_weight(c, c, 3, 1, 1).to(gpu_device).normal_()
        with kernel_log() as log:
            ops.conv_cl(x, w, stride, (1, 0, 0))
        assert log.launches("tconv64_kernel") == want, (B, sorted(log.report))
    tconv(2)
    for (B, T_, Hi, Wi, c, stride) in [(2, 7, 12, 12, 64, (1, 1, 1)), (2, 8, 12, 12, 128, (1, 1, 1)), (2, 8, 12, 12, 64, (2, 1, 1))]:
        x = torch.randn(B, T_, Hi, Wi, c, device=gpu_device)
        w = ops.make_weight(c, c, 3, 1, 1).to(gpu_device).normal_()
        with kernel_log() as log:
            ops.conv_cl(x, w, stride, (1, 0, 0))
        assert log.launches("tconv64_kernel") == 0


@pytest.mark.parametrize("relu", [True, False], ids=["relu", "linear"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_tconv_applies_the_batchnorm_in_front_of_it_bit_identically(shape, relu, gpu_device, kernel_log, tconv):
    """avid_conv_fwd_in / avid_conv_wgrad_in (models/network_blocks.py:36-37, 41-42: spt_bn -> ReLU -> tmp_conv): the layer reads
    the BatchNorm's INPUT and its saved scale / shift and applies fma(x, scale, shift) (+ max(., 0)) while it stages — against the
    same layer reading the tensor avid_bn_fwd_train wrote with that very expression: outputs, BatchNorm partial sums of the output
    and the weight gradient are bit-identical, with and without the residual addend, on ragged tiles (rows past the last
    position must stay zero: ReLU(shift) there would leak into the partial sums) and in clips' padding frames."""
    from avid_hip import lib, ops
    ops._DESC_CACHE.clear()                       # (what a layer offers depends on the tconv switch the fixture set)
    try:
        B, Hi, Wi = shape
        stride, pad = (1, 1, 1), (1, 0, 0)
        x = T(detgen.det_normalish(f"tcaff:{shape}:x", (B, 8, Hi, Wi, 64))).to(gpu_device)
        gamma = T(detgen.det_uniform(f"tcaff:{shape}:gamma", (64,)) * 3.0 - 1.0).to(gpu_device)     # some negative scales
        beta = T(detgen.det_uniform(f"tcaff:{shape}:beta", (64,)) - 0.5).to(gpu_device)
        rm, rv = torch.zeros(64, device=gpu_device), torch.ones(64, device=gpu_device)
        src = ops.BnSource(None, None, relu)
        z = ops.batch_norm_cl(x, gamma, beta, rm, rv, True, relu=relu, src=src)                      # statistics + apply pass
        scale, shift = src.stats4[2].contiguous(), src.stats4[3].contiguous()
        w = ops.make_weight(64, 64, 3, 1, 1)
        w.copy_(T(detgen.det_param(f"tcaff:{shape}:w.weight", (64, 64, 3, 1, 1))))
        w = w.to(gpu_device)
        d = ops._desc_cached((B, 8, Hi, Wi), 64, 64, (3, 1, 1), stride, pad, False)[0]
        assert d.in_affine
        count = lib.raw("avid_debug_in_affine_launches")
        add = T(detgen.det_uniform(f"tcaff:{shape}:add", (B, 8, Hi, Wi, 64))).to(gpu_device)
        for addend in (None, add):
            y_ref, p_ref = ops.conv_cl(z, w, stride, pad, addend=addend, bn_stats=True)
            before = count(1), count(0)
            with kernel_log() as log:
                y, part = ops.conv_fwd_in(x, w, stride, pad, scale, shift, relu=relu, addend=addend, bn_stats=True)
            assert (count(1) - before[0], count(0) - before[1]) == (1, 0) and log.launches("tconv64_kernel<0>") == 1
            assert torch.equal(y, y_ref) and torch.equal(part, p_ref)
        dy = T(detgen.det_uniform(f"tcaff:{shape}:dy", (B, 8, Hi, Wi, 64)) - 0.5).to(gpu_device)
        wg = w.clone().requires_grad_(True)
        with kernel_log() as log:
            ops.conv_cl(z, wg, stride, pad).backward(dy)
        assert log.launches("twgrad64_kernel") == 1
        before = count(1)
        dw = ops.conv_wgrad_in(x, dy, w, stride, pad, scale, shift, relu=relu)
        assert count(1) - before == 1
        assert torch.equal(dw, wg.grad)
    finally:
        ops._DESC_CACHE.clear()


def test_layers_that_cannot_apply_a_batchnorm_say_so(gpu_device, tconv):
    """avid_conv_takes_in_affine is 0 for everything but conv2x's temporal layers, and avid_conv_fwd_in / avid_conv_wgrad_in refuse
    (AVID_E_UNSUPPORTED through the error convention) instead of reading the un-normalised tensor."""
    from avid_hip import lib, ops
    ops._DESC_CACHE.clear()
    try:
        x = torch.randn(2, 4, 14, 14, 128, device=gpu_device)
        w = ops.make_weight(128, 128, 3, 1, 1).normal_().to(gpu_device)
        d = ops._desc_cached((2, 4, 14, 14), 128, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), False)[0]
        assert not d.in_affine
        sc, sh = torch.ones(128, device=gpu_device), torch.zeros(128, device=gpu_device)
        with pytest.raises(Exception, match="does not apply its input's BatchNorm"):
            ops.conv_fwd_in(x, w, (1, 1, 1), (1, 0, 0), sc, sh)
        with pytest.raises(Exception, match="does not apply its input's BatchNorm"):
            ops.conv_wgrad_in(x, torch.randn_like(x), w, (1, 1, 1), (1, 0, 0), sc, sh)
    finally:
        ops._DESC_CACHE.clear()
