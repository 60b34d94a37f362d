"""The step engine and the hardening around it on a real MI355X: optimizer / sampler checkpointing in
torch.optim.Adam's format, the device-resident learning rate, a whole-step hipGraph with RCCL collectives inside,
the BatchNorm hand-over's single-consumer check, eval-mode BatchNorm backward, and the device-side index-error
flag (the reference raises IndexError on an id outside the bank, criterions/avid.py:57-62)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import detgen

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _model(dev):
    import models
    m = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128])
    sd = m.state_dict()
    m.load_state_dict({k: T(detgen.det_param(f"w:{k}", tuple(v.shape)).copy()).to(v.dtype) for k, v in sd.items()})
    return m.to(dev).train()


def _make(dev, N=5000, K=256):
    import criterions
    from avid_hip.parallel import TrainStep
    m = _model(dev)
    crit = criterions.AVID(num_data=N, embedding_dim=128, num_negatives=K, momentum=0.5, device=dev.index)
    gg = torch.Generator().manual_seed(3)
    crit.nce_average.view1_mem.copy_(F.normalize(torch.randn(N, 128, generator=gg), dim=1))
    crit.nce_average.view2_mem.copy_(F.normalize(torch.randn(N, 128, generator=gg), dim=1))
    crit.nce_average.multinomial.reseed(11, 0)
    return m, crit, TrainStep(m, crit)


def _data(dev, N=5000, bs=4, steps=6):
    g = torch.Generator().manual_seed(5)
    video = torch.randn(bs, 3, 8, 64, 64, generator=g).to(dev)
    audio = torch.randn(bs, 1, 40, 100, generator=g).to(dev)
    ids = torch.stack([torch.randperm(N, generator=g)[:bs] for _ in range(steps)]).to(dev)
    return video, audio, ids


def test_trainstep_checkpoint_resume(gpu_device):
    """main-avid.py:115,127,138 save / restore model, criterion and optimizer; a run resumed from such a checkpoint
    continues bit for bit (Adam moments + step, the negative sampler's stream position), and the optimizer state
    is in torch.optim.Adam's own format."""
    video, audio, ids = _data(gpu_device)
    m1, c1, e1 = _make(gpu_device)
    ref = [float(e1.step(video, audio, ids[i])) for i in range(5)]
    m2, c2, e2 = _make(gpu_device)
    got = [float(e2.step(video, audio, ids[i])) for i in range(3)]
    ckp = {"model": {k: v.clone() for k, v in m2.state_dict().items()},
           "criterion": {k: v.clone() for k, v in c2.state_dict().items()}, "optimizer": e2.state_dict()}
    sd = ckp["optimizer"]
    assert set(sd) >= {"state", "param_groups"} and len(sd["state"]) == 141
    assert float(sd["state"][0]["step"]) == 3.0 and sd["avid_sampler"]["offset"] == 3
    for k, p in enumerate(m2.parameters()):
        assert sd["state"][k]["exp_avg"].shape == p.shape
    # torch.optim.Adam accepts it (and what it gives back loads here): the reference's CheckpointManager path
    params = [torch.nn.Parameter(p.detach().clone()) for p in m2.parameters()]
    opt = torch.optim.Adam(params, lr=1e-3)
    opt.load_state_dict({"state": sd["state"], "param_groups": sd["param_groups"]})
    assert opt.param_groups[0]["lr"] == 2e-4 and opt.param_groups[0]["weight_decay"] == 1e-5
    back = opt.state_dict()
    m3, c3, e3 = _make(gpu_device)                      # a fresh process would start like this
    m3.load_state_dict(ckp["model"])
    c3.load_state_dict(ckp["criterion"])
    back["avid_sampler"] = sd["avid_sampler"]
    e3.load_state_dict(back)
    got += [float(e3.step(video, audio, ids[i])) for i in range(3, 5)]
    assert got == ref, (got, ref)
    assert torch.equal(e3.flat.flat, e1.flat.flat) and torch.equal(e3.m, e1.m) and torch.equal(e3.v, e1.v)
    assert int(e3.t_dev) == 5 and e3.t == 5


def test_learning_rate_reaches_a_captured_graph(gpu_device):
    video, audio, ids = _data(gpu_device)
    m, c, e = _make(gpu_device)
    for i in range(2):
        e.step(video, audio, ids[i])
    e.capture(video, audio, ids[2])
    assert e.t == 2 and int(e.t_dev) == 2 and c.nce_average.multinomial.offset == 2    # the capture ran nothing
    e.replay(index=ids[2])
    before = e.flat.flat.clone()
    e.set_lr(0.0)                                       # a scheduler's write: by-value arguments are frozen in the graph
    e.replay(index=ids[3])
    assert torch.equal(e.flat.flat, before)
    e.set_lr(2e-4)
    e.replay(index=ids[4])
    assert not torch.equal(e.flat.flat, before)
    assert e.t == 5 and int(e.t_dev) == 5 and int(c.nce_average.multinomial.offset_dev) == 5


def test_graph_replay_with_rccl_in_the_loop(gpu_device):
    """The whole step INCLUDING its collectives (bucketed gradient all-reduce, fused bank all-gather) captured
    in one hipGraph on a 1-rank RCCL group (AVID_FORCE_DIST=1) replays bit-identically to the eager steps — the
    multi-GPU path then costs one host call per step whatever the host's speed."""
    import torch.distributed as dist
    video, audio, ids = _data(gpu_device)
    m1, c1, e1 = _make(gpu_device)
    eager = [float(e1.step(video, audio, ids[i])) for i in range(6)]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", AVID_FORCE_DIST="1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu_device)
    try:
        m2, c2, e2 = _make(gpu_device)
        assert e2.buckets.comm
        got = [float(e2.step(video, audio, ids[i])) for i in range(3)]
        e2.capture(video, audio, ids[3])
        got += [float(e2.replay(index=ids[i])) for i in range(3, 6)]
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
        os.environ.pop("AVID_FORCE_DIST")
    assert got == eager, (got, eager)
    assert torch.equal(m1.video_model.conv1[0].weight, m2.video_model.conv1[0].weight)
    assert torch.equal(c1.nce_average.view1_mem, c2.nce_average.view1_mem)


def test_bn_handover_checks_for_a_second_consumer(gpu_device):
    """ops.BnSource hands a BatchNorm's backward partial sums over from the next convolution's dgrad — valid only
    if that dgrad's output is the BatchNorm output's WHOLE gradient.  A forward hook that puts an intermediate
    activation into the loss gives it a second consumer: the hand-over must then be dropped for that layer (the
    BatchNorm's backward sees a summed gradient tensor it did not get from the dgrad).  The other BatchNorms of the
    block keep their hand-over, so the comparison with a run that has it disabled everywhere is to fp32
    summation-order noise (1e-5 of scale); partial sums that missed the hook's branch would be off by O(1)."""
    from avid_hip import ops
    from models.network_blocks import BasicR2P1DBlock
    blk = BasicR2P1DBlock(64, 64)
    sd = blk.state_dict()
    blk.load_state_dict({k: T(detgen.det_param(f"h:{k}", tuple(v.shape)).copy()).to(v.dtype) for k, v in sd.items()})
    blk = blk.to(gpu_device).train()
    x = T(detgen.det_normalish("h:x", (2, 4, 12, 12, 64))).to(gpu_device)
    taps = []
    hook = blk.spt_bn1.register_forward_hook(lambda mod, inp, out: taps.append(out))

    def run():
        taps.clear()
        for p in blk.parameters():
            p.grad = None
        xx = x.clone().requires_grad_(True)
        y = blk(xx)
        ((y ** 2).sum() + (taps[0] ** 2).sum() * 0.37).backward()
        return [p.grad.clone() for p in blk.parameters()] + [xx.grad.clone()]

    with_handover = run()
    saved, ops.FUSE_BN_BWD = ops.FUSE_BN_BWD, False
    try:
        without = run()
    finally:
        ops.FUSE_BN_BWD = saved
        hook.remove()
    for a, b in zip(with_handover, without):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), float((a - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("relu", [False, True])
def test_eval_mode_batchnorm_backward(relu, gpu_device):
    """Fine-tuning with frozen BatchNorm: backward through an EVAL-mode BatchNorm (+ReLU) vs float64 torch."""
    from avid_hip import ops
    M, C = 3 * 5 * 7, 128
    x = T(detgen.det_normalish("ebn:x", (M, C))) * 1.3 + 0.2
    g, b = T(detgen.det_param("ebn:bn.weight", (C,))), T(detgen.det_param("ebn:bn.bias", (C,)))
    rm, rv = T(detgen.det_param("ebn:bn.running_mean", (C,))), T(detgen.det_param("ebn:bn.running_var", (C,)))
    gy = T(detgen.det_uniform("ebn:gy", (M, C)))
    xd, gd, bd = (t.to(gpu_device).requires_grad_(True) for t in (x, g, b))
    rmd, rvd = rm.to(gpu_device), rv.to(gpu_device)
    y = ops.batch_norm_cl(xd.view(1, 3, 5, 7, C), gd, bd, rmd, rvd, False, 0.1, 1e-5, relu)
    y.backward(gy.to(gpu_device).view(1, 3, 5, 7, C))
    assert torch.equal(rmd.cpu(), rm) and torch.equal(rvd.cpu(), rv)          # eval mode: buffers untouched
    xr, gr, br = (t.double().requires_grad_(True) for t in (x, g, b))
    yr = F.batch_norm(xr, rm.double(), rv.double(), gr, br, False, 0.1, 1e-5)
    if relu:
        mask = (y.detach().view(M, C) > 0).cpu()
        flips = mask != (yr.detach() > 0)
        assert int(flips.sum()) <= 2 and (not flips.any() or float(yr.detach().abs()[flips].max()) < 1e-5)
        yr = yr * mask.double()
    (yr * gy.double()).sum().backward()

    def rel(a, r):
        return float((a.double().cpu() - r).abs().max() / (r.abs().max() + 1e-30))

    assert rel(y.detach().view(M, C), yr.detach()) < 1e-5
    assert rel(xd.grad.view(M, C), xr.grad) < 1e-5 and rel(gd.grad, gr.grad) < 2e-5 and rel(bd.grad, br.grad) < 2e-5


def test_out_of_range_index_raises(gpu_device):
    """criterions/avid.py:57-62,124: an id outside [0, num_data) raises in the reference.  The kernels flag it in a
    device word (and stay inside the table); the criterion's next forward — or ops.check_device_errors() —
    raises IndexError.  Nothing synchronises on the clean path."""
    import criterions
    from avid_hip import ops
    N, bs = 1000, 4
    crit = criterions.AVID(num_data=N, embedding_dim=128, num_negatives=32, momentum=0.5, device=gpu_device.index)
    v = torch.randn(bs, 128, device=gpu_device, requires_grad=True)
    a = torch.randn(bs, 128, device=gpu_device, requires_grad=True)
    good = torch.tensor([1, 5, 9, 200], device=gpu_device)
    ops.check_device_errors(gpu_device)
    crit(v, a, good)[0].backward()
    ops.check_device_errors(gpu_device)                                       # clean
    bank = crit.nce_average.view1_mem.clone()
    bad = torch.tensor([1, 5, N + 7, 200], device=gpu_device)
    loss, _ = crit(v, a, bad)                                                 # runs (clamped), flags
    assert torch.isfinite(loss)
    with pytest.raises(IndexError, match="outside"):
        ops.check_device_errors(gpu_device)
    rows = torch.tensor([1, 5, 200], device=gpu_device)
    assert not torch.equal(crit.nce_average.view1_mem[rows], bank[rows])      # valid samples were still updated
    ops.check_device_errors(gpu_device)                                       # the flag was cleared by the raise
    # the non-blocking poll: flagged in one forward, raised by a later one
    crit(v, a, torch.tensor([-3, 5, 9, 200], device=gpu_device))
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        for _ in range(3):                       # first poll starts the copy, a later one sees it
            crit(v, a, good)
            torch.cuda.synchronize()
    crit(v, a, good)
    ops.check_device_errors(gpu_device)


@pytest.mark.parametrize("cin,cout,shape", [(64, 128, (2, 4, 10, 12)), (64, 128, (3, 5, 9, 11)), (128, 256, (4, 4, 14, 14)),
                                            (256, 512, (8, 2, 7, 7))])
def test_residual_conv_compact_gradient(cin, cout, shape, gpu_device):
    """Stage-transition blocks (models/network_blocks.py:47-51,58): the 1x1x1 / (2,2,2) residual convolution is
    computed inside spt_conv1's op; its input gradient stays on the sub-sampled grid (a dense 1x1x1 dgrad) and is
    added inside spt_conv1's strided dgrad as a compact addend — directly written parity classes and K-split ones
    (the reduce) alike, odd extents included.  Same forward bit for bit, same gradients to fp32 summation order as the
    path that scatters it into an x-shaped tensor first (which test_blocks_vs_reference_golden pins to the reference)."""
    import models.network_blocks as nb
    blk = nb.BasicR2P1DBlock(cin, cout, stride=(2, 2, 2))
    sd = blk.state_dict()
    blk.load_state_dict({k: T(detgen.det_param(f"rc:{k}", tuple(v.shape)).copy()).to(v.dtype) for k, v in sd.items()})
    blk = blk.to(gpu_device).train()
    B, Tt, H, W = shape
    x = T(detgen.det_normalish(f"rc:x:{shape}:{cin}", (B, Tt, H, W, cin))).to(gpu_device)

    def run(fused):
        saved, nb._FUSE_RES = nb._FUSE_RES, fused
        try:
            for p in blk.parameters():
                p.grad = None
            xx = x.clone().requires_grad_(True)
            y = blk(xx)
            g = T(detgen.det_uniform(f"rc:g:{shape}:{cout}", tuple(y.shape))).to(gpu_device)
            y.backward(g)
            return y.detach().clone(), [xx.grad.clone()] + [p.grad.clone() for p in blk.parameters()]
        finally:
            nb._FUSE_RES = saved

    y1, g1 = run(True)
    y0, g0 = run(False)
    assert torch.equal(y1, y0)
    for a, b in zip(g1, g0):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-12, float((a - b).abs().max() / b.abs().max())


def test_trailing_wgrad_streams_change_nothing(gpu_device, per_layer_path):
    """The weight gradients run on helper streams that trail the backward chain and are joined before Adam
    (ops.deferred_wgrads, on by default under TrainStep): same kernels, same order of summation — three steps with
    and without give identical losses, parameters and Adam moments, with the two towers on two streams as well."""
    from avid_hip import ops
    video, audio, ids = _data(gpu_device)
    res = {}
    keep = ops.DEFER_WGRAD
    try:
        for on in (0, 1):
            ops.DEFER_WGRAD = on
            m, c, e = _make(gpu_device)
            losses = [float(e.step(video, audio, ids[i])) for i in range(3)]
            torch.cuda.synchronize()
            assert bool(ops._DEFERRED) or not on           # the helper streams were really used
            res[on] = (losses, e.flat.flat.clone(), e.m.clone(), e.v.clone())
    finally:
        ops.DEFER_WGRAD = keep
    assert res[0][0] == res[1][0]
    for a, b in zip(res[0][1:], res[1][1:]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("helper_stream", [True, False])
@pytest.mark.parametrize("variant", ["wino", "wino2"])
def test_winograd_weight_tables_change_nothing(variant, helper_stream, gpu_device, monkeypatch, per_layer_path):
    """From its second step on the per-layer step engine keeps the Winograd transforms of the weights for the INPUT
    GRADIENTS current with ONE launch per step (ops.TransposedWeights.refresh_wino, on the helper stream) instead of a
    transform launch inside every input-gradient call; the forward's transforms stay inside the calls.  Same kernels,
    same arithmetic: losses and parameters after four steps must be IDENTICAL to the run that transforms inside every
    call (table off) — and the table must really have been used."""
    from avid_hip import ops, lib
    ops.wino_configure(1, 1, 128)
    if variant == "wino2":
        ops.wino2_configure(0)
    if not helper_stream:        # no helper stream: the tables refreshed on the main stream behind the forward
        monkeypatch.setattr(ops, "DEFER_WGRAD", 0)
    try:
        outs = []
        for table in (False, True):
            m, crit, ts = _make(gpu_device)
            ts.twt.wino_on = table
            video, audio, ids = _data(gpu_device, steps=4)
            losses = [float(ts.step(video, audio, ids[i])) for i in range(3)]
            lib.timing_enable(True)
            losses.append(float(ts.step(video, audio, ids[3])))
            torch.cuda.synchronize()
            rep = lib.timing_report()
            lib.timing_enable(False)
            n_weight = rep.get("wino_weight_kernel", {"launches": 0})["launches"]
            n_wino = sum(v["launches"] for k, v in rep.items() if k.startswith(("wino2_kernel<", "wino2p_kernel<") if variant == "wino2" else "wino_kernel<"))
            assert n_wino >= 18                                   # 9 layers, forward and input gradient
            if table:             # input gradients from the table, the forward transforms in the call
                assert ts.twt.n_wino == n_wino // 2 and n_weight == n_wino // 2, (ts.twt.n_wino, n_wino, n_weight)
            else:
                assert n_weight == n_wino
            outs.append((losses, [p.detach().clone() for p in m.parameters()]))
        for other in outs[1:]:
            assert outs[0][0] == other[0]
            for a, b in zip(outs[0][1], other[1]):
                assert torch.equal(a, b)
    finally:
        ops.wino_configure(-1, -1, -1)
        ops.wino2_configure(-1)
