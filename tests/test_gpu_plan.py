"""The compiled launch programs (avid_hip/plan.py + csrc/program.hip: the whole model forward / backward as one C call
each) against the per-layer path (one autograd.Function per layer, avid_hip/ops.py) on a real MI355X.

Both paths call the same entry points with the same arguments, so everything must agree BIT FOR BIT: embeddings,
loss, every parameter gradient, the BatchNorm running statistics — inside the step engine (gradients written into the
flat buffer, grouped weight gradients, trailing streams) and in the reference's own loop shape (main-avid.py:155-180:
``model()`` -> ``criterion()`` -> ``loss.item()`` -> ``zero_grad / backward / step`` with torch.optim.Adam)."""
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


def _crit(dev, N=5000, K=256):
    import criterions
    crit = criterions.AVID(num_data=N, embedding_dim=128, num_negatives=K, momentum=0.5, device=dev.index)
    gg = torch.Generator().manual_seed(3)
    crit.nce_average.view1_mem.copy_(F.normalize(torch.randn(N, 128, generator=gg), dim=1))
    crit.nce_average.view2_mem.copy_(F.normalize(torch.randn(N, 128, generator=gg), dim=1))
    crit.nce_average.multinomial.reseed(11, 0)
    return crit


def _data(dev, N=5000, bs=4, steps=4, hw=64):
    g = torch.Generator().manual_seed(5)
    video = torch.randn(bs, 3, 8, hw, hw, generator=g).to(dev)
    audio = torch.randn(bs, 1, 40, 100, generator=g).to(dev)
    ids = torch.stack([torch.randperm(N, generator=g)[:bs] for _ in range(steps)]).to(dev)
    return video, audio, ids


class _plan_switch:
    def __init__(self, on):
        self.on = on

    def __enter__(self):
        from avid_hip import plan
        self.prev, plan.ENABLED = plan.ENABLED, self.on

    def __exit__(self, *exc):
        from avid_hip import plan
        plan.ENABLED = self.prev


def _engine_steps(dev, plan_on, steps, bs, hw):
    from avid_hip.parallel import TrainStep
    with _plan_switch(plan_on):
        m, crit = _model(dev), _crit(dev)
        eng = TrainStep(m, crit)
        video, audio, ids = _data(dev, bs=bs, steps=steps, hw=hw)
        losses, grads = [], None
        for i in range(steps):
            loss = eng.forward_backward(video, audio, ids[i])
            if i == steps - 1:
                torch.cuda.synchronize()
                grads = eng.flat.grad.clone()
            eng.optimizer_step()
            losses.append(float(loss))
        torch.cuda.synchronize()
        return losses, grads, {k: v.clone() for k, v in m.state_dict().items()}, eng


@pytest.mark.parametrize("bs,hw", [(4, 64), (16, 112)])
def test_engine_step_through_programs_is_bit_identical(gpu_device, bs, hw):
    """TrainStep over the launch programs == TrainStep over the per-layer autograd Functions: loss of every step, the
    flat gradient buffer of the last step, all parameters and BatchNorm buffers after the last Adam step."""
    from avid_hip import plan
    l0, g0, sd0, _ = _engine_steps(gpu_device, False, 3, bs, hw)
    l1, g1, sd1, eng = _engine_steps(gpu_device, True, 3, bs, hw)
    pls = [p for p in eng.model.__dict__.get("_avid_plans", {}).values() if p]
    assert pls, "the step did not run through a launch program"
    assert l0 == l1
    assert torch.equal(g0, g1)
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k


def test_overlapped_optimizer_step_changes_nothing(gpu_device):
    """TrainStep.step() updates everything but the video stem's three parameters on the fourth stream, beside the stem's
    weight gradient, and the stem's afterwards (parallel._optimizer_step_overlapped): parameters, Adam moments and the
    step counter after three steps are bit-identical to forward_backward() + optimizer_step()."""
    from avid_hip.parallel import TrainStep
    dev = gpu_device
    video, audio, ids = _data(dev, bs=4, steps=3, hw=64)
    out = []
    for fused in (True, False):
        m, crit = _model(dev), _crit(dev)
        eng = TrainStep(m, crit)
        used = []
        for i in range(3):
            if fused:
                eng.step(video, audio, ids[i])
                used.append(eng._step_plan is not None and bool(eng._step_plan.adam_early))
            else:
                eng.forward_backward(video, audio, ids[i])
                eng.optimizer_step()
        torch.cuda.synchronize()
        if fused:
            assert all(used), "step() did not take the overlapped optimizer path"
        out.append((eng.flat.flat.clone(), eng.m.clone(), eng.v.clone(), int(eng.t_dev) if eng.t_dev is not None else eng.t))
    for a, b in zip(out[0][:3], out[1][:3]):
        assert torch.equal(a, b)
    assert out[0][3] == out[1][3] == 3


def test_reference_loop_through_programs(gpu_device):
    """main-avid.py:155-180 as written — model(), criterion(), loss.item(), zero_grad, backward, torch.optim.Adam.step —
    takes the program path (one autograd node for the model) and ends where the step engine ends: same losses, and the
    gradients handed to autograd are bit-identical to the ones the engine writes into its flat buffer."""
    from avid_hip import plan
    dev = gpu_device
    steps = 3
    l_eng, g_eng, _, eng = _engine_steps(dev, True, steps, 4, 64)
    m, crit = _model(dev), _crit(dev)
    opt = torch.optim.Adam(m.parameters(), lr=2e-4, betas=(0.9, 0.999), weight_decay=1e-5)
    video, audio, ids = _data(dev, bs=4, steps=steps, hw=64)
    losses = []
    for i in range(steps):
        v, a = m(video, audio)
        assert type(v.grad_fn).__name__.startswith("NetFn"), "the model call did not go through the launch program"
        loss, _ = crit(v, a, ids[i])
        losses.append(loss.item())
        opt.zero_grad()
        loss.backward()
        if i == steps - 1:
            got = {id(p): p.grad.clone() for p in m.parameters()}
        opt.step()
    # step 0 is bit-identical (same weights); later steps differ by torch.optim.Adam's arithmetic vs the fused kernel's
    assert losses[0] == l_eng[0]
    # (third step: weights two Adam steps apart in rounding — 1e-6 of a weight on 4 clips moves the loss by 2e-5 .. 3e-5)
    np.testing.assert_allclose(losses, l_eng, rtol=1e-4)
    # gradients of the last step against the engine's flat buffer: the same kernels on weights that differ by two Adam
    # steps' rounding (torch.optim.Adam vs the fused kernel) — layout identical, values close in norm
    flat = eng.flat
    num = den = 0.0
    for p_ref, (p, o) in zip(reversed(list(m.parameters())), zip(flat.params, flat.offsets)):
        want = g_eng[o:o + p.numel()].as_strided(p.shape, p.stride())
        gotp = got[id(p_ref)]
        assert gotp.shape == want.shape and gotp.stride() == want.stride()
        num += float((gotp - want).double().pow(2).sum())
        den += float(want.double().pow(2).sum())
    assert (num / den) ** 0.5 < 0.15, (num / den) ** 0.5


def test_first_step_gradients_equal_the_engines(gpu_device):
    """Same weights, same batch: the gradients the program hands to autograd (reference loop) are bit-identical to the
    ones it writes into the engine's flat buffer, and to the per-layer path's."""
    from avid_hip.parallel import TrainStep
    dev = gpu_device
    video, audio, ids = _data(dev, bs=4, steps=1, hw=64)
    m, crit = _model(dev), _crit(dev)
    v, a = m(video, audio)
    loss, _ = crit(v, a, ids[0])
    loss.backward()
    torch.cuda.synchronize()
    got = [p.grad.clone() for p in m.parameters()]
    for plan_on in (True, False):
        with _plan_switch(plan_on):
            m2, crit2 = _model(dev), _crit(dev)
            eng = TrainStep(m2, crit2)
            loss2 = eng.forward_backward(video, audio, ids[0])
            torch.cuda.synchronize()
            assert float(loss2) == float(loss)
            for g, p in zip(got, m2.parameters()):
                assert torch.equal(g, p.grad), (plan_on, tuple(p.shape))


def test_hooked_or_eval_model_takes_the_per_layer_path(gpu_device):
    """Forward hooks (the oracle's ReLU-mask capture), eval mode and no_grad bypass the programs."""
    dev = gpu_device
    video, audio, _ = _data(dev, bs=2, steps=1, hw=64)
    m = _model(dev)
    v, _ = m(video, audio)
    assert type(v.grad_fn).__name__.startswith("NetFn")
    h = m.video_model.conv2x[0].spt_bn1.register_forward_hook(lambda mod, i, o: None)
    v, _ = m(video, audio)
    assert not type(v.grad_fn).__name__.startswith("NetFn")
    h.remove()
    with torch.no_grad():
        v, _ = m(video, audio)
    assert v.grad_fn is None
    m.eval()
    v, _ = m(video, audio)
    assert not type(v.grad_fn).__name__.startswith("NetFn")


def test_a_stale_program_is_not_reused(gpu_device):
    """What a compiled Plan froze besides shapes is part of its cache key: a parameter frozen AFTER the first
    step sends the next call to the per-layer path (no gradient for the frozen parameter, every other gradient as before),
    un-freezing it brings the program back; BatchNorm buffers re-created by `.cpu()` / `.cuda()` get a fresh program whose
    running statistics keep updating."""
    dev = gpu_device
    video, audio, _ = _data(dev, bs=2, steps=1, hw=64)
    m = _model(dev)

    def step():
        for p in m.parameters():
            p.grad = None
        v, a = m(video, audio)
        (v.square().sum() + a.square().sum()).backward()
        return v

    v = step()
    assert type(v.grad_fn).__name__.startswith("NetFn")
    ref = {n: p.grad.clone() for n, p in m.named_parameters()}
    w = m.video_model.conv3x[0].tmp_conv1.weight
    w.requires_grad_(False)
    v = step()
    assert not type(v.grad_fn).__name__.startswith("NetFn") and w.grad is None
    for n, p in m.named_parameters():
        if p is not w:
            assert float((p.grad - ref[n]).abs().max()) <= 1e-4 * float(ref[n].abs().max()) + 1e-12, n
    w.requires_grad_(True)
    v = step()
    assert type(v.grad_fn).__name__.startswith("NetFn") and w.grad is not None
    m.cpu()
    m.to(dev)
    bn = m.video_model.conv2x[0].spt_bn1
    before = bn.running_mean.clone()
    v = step()
    torch.cuda.synchronize()
    assert type(v.grad_fn).__name__.startswith("NetFn")
    assert not torch.equal(bn.running_mean, before)         # the new buffers are the ones the program updates


def test_backward_twice_raises(gpu_device):
    dev = gpu_device
    video, audio, _ = _data(dev, bs=2, steps=1, hw=64)
    m = _model(dev)
    v, a = m(video, audio)
    (v.sum() + a.sum()).backward(retain_graph=True)
    with pytest.raises(RuntimeError):
        (v.sum() + a.sum()).backward()


def test_placed_streams_run_concurrently(gpu_device):
    """avid_hip/streams.py: the audio tower's, the trailing and the collectives' stream of the current stream overlap with
    it and with each other (single-wave probe kernels: a serialised pair takes twice as long) — also when other streams
    were created first, which is what moves hardware queues onto shared dispatch pipes."""
    from avid_hip import streams
    clutter = [torch.cuda.Stream(gpu_device) for _ in range(3)]        # (an RCCL communicator, a test, a data loader ...)
    s = torch.cuda.Stream(gpu_device)
    with torch.cuda.stream(s):
        ss = streams.place(gpu_device)
        assert streams.place(gpu_device) is ss                           # cached per compute stream
        with torch.cuda.stream(ss.side):
            assert streams.current_set(gpu_device) is ss                 # the audio tower's stream finds its set
    assert ss.report["probed"] and ss.report["independent"] == 4, ss.report
    four = [ss.main, ss.side, ss.trail, ss.comm]
    alone = min(streams._alone_us(ss.main) for _ in range(3))
    for a in range(4):
        for b in range(a + 1, 4):
            assert streams.concurrent(four[a], four[b], alone), (a, b, ss.report)
    del clutter
