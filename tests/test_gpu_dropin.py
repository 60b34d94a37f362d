"""The reference's own loop (main-avid.py:155-180) with the two objects its factories build swapped for this build's:
`avid_hip.parallel.DistributedDataParallel` for utils/main_utils.py:112 and `avid_hip.parallel.Adam` for :250.  The loop body is
the reference's, statement for statement; what it reaches is the step engine's result BIT FOR BIT (same launch programs, same
flat buffers, same Adam kernel), and each object also works beside torch's version of the other one."""
import os
import socket

import numpy as np
import pytest
import torch

from test_gpu_plan import _model, _crit, _data, _engine_steps, _plan_switch

pytestmark = pytest.mark.gpu


def _loop(net, crit, opt, video, audio, ids, steps, sched=None):
    """main-avid.py:169-178."""
    losses = []
    for i in range(steps):
        v, a = net(video, audio)
        loss, _ = crit(v, a, ids[i])
        losses.append(loss.item())
        opt.zero_grad()
        loss.backward()
        opt.step()
        if sched is not None:
            sched.step()
    torch.cuda.synchronize()
    return losses


class _one_rank_group:
    """A one-rank RCCL group with AVID_FORCE_DIST=1: the bucketed collectives really run."""

    def __enter__(self):
        import torch.distributed as dist
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        self.prev = os.environ.get("AVID_FORCE_DIST")
        os.environ["AVID_FORCE_DIST"] = "1"
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))

    def __exit__(self, *exc):
        import torch.distributed as dist
        dist.destroy_process_group()
        os.environ.pop("AVID_FORCE_DIST", None)
        if self.prev is not None:
            os.environ["AVID_FORCE_DIST"] = self.prev


@pytest.mark.parametrize("rccl", [False, True], ids=["no_group", "one_rank_rccl"])
def test_reference_loop_with_both_dropins_is_the_engine_bit_for_bit(gpu_device, rccl):
    from avid_hip import parallel
    dev, steps = gpu_device, 3
    l_eng, g_eng, sd_eng, eng = _engine_steps(dev, True, steps, 4, 64)
    opt_eng = eng.state_dict()

    def run():
        m, crit = _model(dev), _crit(dev)
        net = parallel.DistributedDataParallel(m, device_ids=[dev.index])
        opt = parallel.Adam(net.parameters(), lr=2e-4, weight_decay=1e-5, betas=[0.9, 0.999])
        assert opt.flat is net._engine.flat, "the optimizer did not adopt the wrapper's flat buffers"
        assert net._engine.buckets.comm == rccl
        video, audio, ids = _data(dev, bs=4, steps=steps, hw=64)
        losses = _loop(net, crit, opt, video, audio, ids, steps)
        assert [p for p in m.__dict__.get("_avid_plans", {}).values() if p], "the loop did not run through a launch program"
        return m, net, opt, losses

    if rccl:
        with _one_rank_group():
            m, net, opt, losses = run()
    else:
        m, net, opt, losses = run()
    assert losses == l_eng
    sd = m.state_dict()
    for k in sd_eng:
        assert torch.equal(sd[k], sd_eng[k]), k
    assert torch.equal(net._engine.flat.grad, g_eng)
    assert all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(opt.flat.params, opt.flat.grad_views))
    # optimizer state: torch.optim.Adam's format, equal to the engine's
    got = opt.state_dict()
    assert sorted(got["state"]) == sorted(opt_eng["state"]) and len(got["state"]) == len(list(m.parameters()))
    for k, st in opt_eng["state"].items():
        assert float(got["state"][k]["step"]) == float(st["step"]) == steps
        assert torch.equal(got["state"][k]["exp_avg"], st["exp_avg"]) and torch.equal(got["state"][k]["exp_avg_sq"], st["exp_avg_sq"])
    assert sorted(net.state_dict()) == sorted("module." + k for k in sd)
    assert net.module.out_dim == m.out_dim                    # main-avid.py:100


def test_dropin_wrapper_with_torch_adam_and_dropin_adam_with_the_bare_model(gpu_device):
    """Each object beside torch's version of the other.  Wrapper + torch.optim.Adam: `optimizer.zero_grad()` sets every `.grad` to
    None, the backward program writes the flat buffer, and the end-of-backward callback seats `.grad` again — the first
    step's gradients are the engine's bit for bit, the losses follow torch.optim.Adam's arithmetic.  Adam drop-in + the
    bare model: it flattens the parameters itself; losses follow torch.optim.Adam's, parameters equal the engine's."""
    from avid_hip import parallel
    dev, steps = gpu_device, 3
    video, audio, ids = _data(dev, bs=4, steps=steps, hw=64)
    # reference: the bare model with torch.optim.Adam (tests/test_gpu_plan.py::test_reference_loop_through_programs)
    m0, c0 = _model(dev), _crit(dev)
    l0 = _loop(m0, c0, torch.optim.Adam(m0.parameters(), lr=2e-4, betas=(0.9, 0.999), weight_decay=1e-5), video, audio, ids, steps)
    # (a) wrapper + torch's Adam
    m1, c1 = _model(dev), _crit(dev)
    net = parallel.DistributedDataParallel(m1, device_ids=[dev.index])
    l1 = _loop(net, c1, torch.optim.Adam(net.parameters(), lr=2e-4, betas=(0.9, 0.999), weight_decay=1e-5), video, audio, ids, steps)
    assert all(p.grad is not None for p in m1.parameters())
    assert l1[0] == l0[0]
    np.testing.assert_allclose(l1, l0, rtol=2e-5)
    for (k, a), b in zip(m1.state_dict().items(), m0.state_dict().values()):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6, msg=k)
    # (b) the bare model + the Adam drop-in
    m2, c2 = _model(dev), _crit(dev)
    opt2 = parallel.Adam(m2.parameters(), lr=2e-4, betas=(0.9, 0.999), weight_decay=1e-5)
    assert all(p.data_ptr() == opt2.flat.flat.data_ptr() + 4 * o for p, o in zip(opt2.flat.params, opt2.flat.offsets))
    l2 = _loop(m2, c2, opt2, video, audio, ids, steps)
    assert l2[0] == l0[0]
    np.testing.assert_allclose(l2, l0, rtol=1e-4)
    # the gradients reach the seated `.grad` views through autograd's in-place add onto the zeroed flat buffer (0 + g: exact) and
    # the update is the engine's kernel with the engine's constants: the run is the ENGINE's, bit for bit (against torch.optim.Adam
    # the weights differ by up to lr per step where a gradient is ~eps: no elementwise bar holds there)
    l_eng, _, sd_eng, _ = _engine_steps(dev, True, steps, 4, 64)
    assert l2 == l_eng
    for k, a in m2.state_dict().items():
        assert torch.equal(a, sd_eng[k]), k


def test_dropin_adam_is_a_torch_optimizer(gpu_device):
    """MultiStepLR drives it (utils/main_utils.py:258), its state_dict loads into torch.optim.Adam and torch's into it
    (CheckpointManager, main-avid.py:115,127,138), and a resumed run continues bit for bit."""
    from avid_hip import parallel
    dev = gpu_device
    torch.manual_seed(0)
    ws = [torch.randn(64, 32, device=dev), torch.randn(7, device=dev), torch.randn(16, 8, 3, 3, device=dev)]
    gs = [[torch.randn_like(w) for w in ws] for _ in range(6)]

    def make(cls):
        ps = [torch.nn.Parameter(w.clone()) for w in ws]
        opt = cls(ps, lr=1e-2, betas=(0.9, 0.999), weight_decay=1e-5)
        return ps, opt, torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[2], gamma=0.1)

    def steps(ps, opt, sched, lo, hi):
        for t in range(lo, hi):
            opt.zero_grad()
            for p, g in zip(ps, gs[t]):
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
            opt.step()
            sched.step()

    pt, ot, st = make(torch.optim.Adam)
    po, oo, so = make(parallel.Adam)
    steps(pt, ot, st, 0, 4)
    steps(po, oo, so, 0, 4)
    assert oo.param_groups[0]["lr"] == pytest.approx(1e-3) == ot.param_groups[0]["lr"]
    for a, b in zip(po, pt):
        torch.testing.assert_close(a, b, rtol=2e-5, atol=1e-7)
    # ours -> torch, torch -> ours, ours -> ours (bit for bit)
    sd = oo.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 4
    _, ot2, _ = make(torch.optim.Adam)
    ot2.load_state_dict(sd)
    torch.testing.assert_close(ot2.state_dict()["state"][2]["exp_avg"], ot.state_dict()["state"][2]["exp_avg"], rtol=2e-5, atol=1e-8)
    po2, oo2, so2 = make(parallel.Adam)
    oo2.load_state_dict(ot.state_dict())
    assert oo2._t == 4 and oo2.param_groups[0]["lr"] == pytest.approx(1e-3)
    torch.testing.assert_close(oo2.state_dict()["state"][1]["exp_avg_sq"], sd["state"][1]["exp_avg_sq"], rtol=2e-5, atol=1e-10)
    po3, oo3, so3 = make(parallel.Adam)
    oo3.load_state_dict(sd)
    so3.load_state_dict(so.state_dict())
    for a, b in zip(po3, po):
        a.data.copy_(b.data)
    steps(po, oo, so, 4, 6)
    steps(po3, oo3, so3, 4, 6)
    for a, b in zip(po3, po):
        assert torch.equal(a, b)
    with pytest.raises(NotImplementedError):
        parallel.Adam([torch.nn.Parameter(ws[0].clone())], amsgrad=True)


def test_dropin_wrapper_on_the_per_layer_path(gpu_device):
    """With the launch programs off (AVID_PLAN=0 / a hooked module) the gradients come through autograd: every parameter's hook
    moves its gradient into the flat buffer.  Same kernels per layer, so the first step's loss is the program path's bit
    for bit and the run stays close to it."""
    from avid_hip import parallel
    dev, steps = gpu_device, 2
    l_eng, _, sd_eng, _ = _engine_steps(dev, True, steps, 4, 64)
    with _plan_switch(False):
        m, crit = _model(dev), _crit(dev)
        net = parallel.DistributedDataParallel(m, device_ids=[dev.index])
        opt = parallel.Adam(net.parameters(), lr=2e-4, weight_decay=1e-5)
        video, audio, ids = _data(dev, bs=4, steps=steps, hw=64)
        losses = _loop(net, crit, opt, video, audio, ids, steps)
    assert not [p for p in m.__dict__.get("_avid_plans", {}).values() if p]
    assert losses[0] == l_eng[0]
    np.testing.assert_allclose(losses, l_eng, rtol=1e-4)
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(opt.flat.params, opt.flat.grad_views))


def test_reference_factories_through_utils_main_utils_build_the_dropins(gpu_device, tmp_path, monkeypatch):
    """`utils.main_utils` of this package = the reference's module body with `torch` bound to a proxy (avid-cma_amd/utils/main_utils.py):
    the objects `distribute_model_to_cuda` (utils/main_utils.py:112) and `build_optimizer` (:250) hand to main-avid.py's loop are the
    flat-buffer ones — without an edit to the reference.  The GPU box has no reference checkout, so a stand-in file with the two
    call sites' shape (written here, in this test's own words) plays its part; the real file is checked on the build box
    (tests/test_host_logic.py).  The loop's result is the step engine's bit for bit."""
    import importlib
    import sys
    import types
    from avid_hip import parallel
    ref = tmp_path / "ref"
    (ref / "utils").mkdir(parents=True)
    (ref / "utils" / "main_utils.py").write_text(
        "import torch\n\n"
        "def distribute_model_to_cuda(models, args, batch_size, num_workers, ngpus_per_node):\n"
        "    torch.cuda.set_device(args.gpu)\n"
        "    models.cuda(args.gpu)\n"
        "    return torch.nn.parallel.DistributedDataParallel(models, device_ids=[args.gpu]), args, batch_size // ngpus_per_node, num_workers\n\n"
        "def build_optimizer(params, cfg, logger=None):\n"
        "    o = torch.optim.Adam(params=params, lr=cfg['lr']['base_lr'], weight_decay=cfg['weight_decay'], betas=cfg['betas'])\n"
        "    return o, torch.optim.lr_scheduler.MultiStepLR(o, milestones=cfg['lr']['milestones'], gamma=cfg['lr']['gamma'])\n")
    import utils
    monkeypatch.setattr(utils, "__path__", list(utils.__path__) + [str(ref / "utils")])
    monkeypatch.delitem(sys.modules, "utils.main_utils", raising=False)
    monkeypatch.delenv("AVID_DROPIN", raising=False)
    mu = importlib.import_module("utils.main_utils")
    try:
        assert mu.REFERENCE_FILE == str(ref / "utils" / "main_utils.py")
        dev, steps = gpu_device, 3
        l_eng, g_eng, sd_eng, _ = _engine_steps(dev, True, steps, 4, 64)
        m, crit = _model(dev), _crit(dev)
        args = types.SimpleNamespace(gpu=dev.index, distributed=True)
        net, _, bs, _ = mu.distribute_model_to_cuda(m, args, 4, 0, 1)
        opt, sched = mu.build_optimizer(list(net.parameters()) + list(crit.parameters()),
                                        {"lr": {"base_lr": 2e-4, "milestones": [100], "gamma": 1.0}, "weight_decay": 1e-5, "betas": [0.9, 0.999]})
        assert isinstance(net, parallel.DistributedDataParallel) and type(opt) is parallel.Adam and bs == 4
        # anything but this build's two-tower model, and any parameter list but the wrapper's, gets torch's own objects
        lin = torch.nn.Linear(4, 4).to(dev)
        assert type(mu.build_optimizer(lin.parameters(), {"lr": {"base_lr": 1e-3, "milestones": [1], "gamma": 1.0}, "weight_decay": 0.0,
                                                          "betas": [0.9, 0.999]})[0]) is torch.optim.Adam
        assert opt.flat is net._engine.flat
        video, audio, ids = _data(dev, bs=4, steps=steps, hw=64)
        losses = _loop(net, crit, opt, video, audio, ids, steps, sched=sched)
        assert losses == l_eng
        sd = m.state_dict()
        for k in sd_eng:
            assert torch.equal(sd[k], sd_eng[k]), k
        assert torch.equal(net._engine.flat.grad, g_eng)
    finally:
        sys.modules.pop("utils.main_utils", None)
