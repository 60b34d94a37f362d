"""World-size-2 gloo tests (CPU) of the multi-GPU path: bucketed overlapped gradient all-reduce over
the flat gradient buffer, the fused bank-update all-gather, and _gather_from_all's rank order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def run2(fn):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, fn, ret)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return dict(ret)


def _tiny_model():
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(),
                            torch.nn.Linear(16, 4))
    return m


def _grad_job(rank, world):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "avid-cma_amd"))
    from avid_hip.parallel import FlatParams, GradBuckets
    m = _tiny_model()
    flat = FlatParams(m)
    buckets = GradBuckets(flat, bucket_bytes=256)          # tiny buckets -> several async all-reduces
    assert len(buckets.bounds) >= 3
    torch.manual_seed(100 + rank)
    x = torch.randn(5, 8)
    out = []
    for _ in range(2):                                     # two steps: hooks re-arm
        flat.zero_grad()
        m(x).pow(2).sum().backward()
        buckets.finish()
        out.append(flat.grad.clone() / world)
    return [o.numpy() for o in out], [p.grad.data_ptr() == flat.grad.data_ptr() + 4 * o
                                      for p, o in zip(flat.params, flat.offsets)]


def test_bucketed_allreduce_equals_mean_of_rank_grads():
    res = run2(_grad_job)
    ref = []
    for rank in range(2):
        m = _tiny_model()
        torch.manual_seed(100 + rank)
        x = torch.randn(5, 8)
        m(x).pow(2).sum().backward()
        ref.append([p.grad.clone() for p in reversed(list(m.parameters()))])
    for rank in range(2):
        grads, seated = res[rank]
        assert all(seated)
        flat = torch.from_numpy(grads[0])
        off = 0
        for g0, g1 in zip(ref[0], ref[1]):
            n = g0.numel()
            torch.testing.assert_close(flat[off:off + n].view_as(g0), (g0 + g1) / 2, rtol=1e-6, atol=1e-7)
            off += (n + 3) // 4 * 4
        torch.testing.assert_close(torch.from_numpy(grads[1]), flat, rtol=1e-6, atol=1e-7)   # step 2 == step 1
    torch.testing.assert_close(torch.from_numpy(res[0][0][0]), torch.from_numpy(res[1][0][0]))


def _gather_job(rank, world):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "avid-cma_amd"))
    from criterions.avid import gather_update_records
    from utils.distributed_utils import _gather_from_all
    torch.manual_seed(rank)
    v, a = torch.randn(3, 128), torch.randn(3, 128)
    y = torch.tensor([10 + rank, 2 ** 40 + rank, 7], dtype=torch.int64)      # ids beyond 2^32 survive the bit-cast
    va, aa, ya = gather_update_records(v, a, y)
    g = _gather_from_all(torch.full((2, 2), float(rank)))
    return va.numpy(), aa.numpy(), ya.numpy(), g.numpy()


def test_fused_bank_allgather_and_rank_order():
    res = run2(_gather_job)
    vs, as_, ys = [], [], []
    for rank in range(2):
        torch.manual_seed(rank)
        vs.append(torch.randn(3, 128)); as_.append(torch.randn(3, 128))
        ys.append(torch.tensor([10 + rank, 2 ** 40 + rank, 7], dtype=torch.int64))
    for rank in range(2):
        va, aa, ya, g = res[rank]
        assert (torch.from_numpy(va) == torch.cat(vs)).all() and (torch.from_numpy(aa) == torch.cat(as_)).all()
        assert ya.tolist() == torch.cat(ys).tolist()
        assert g.tolist() == [[0, 0], [0, 0], [1, 1], [1, 1]]


def _cma_shard_job(rank, world):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "avid-cma_amd"))
    from criterions.avid_cma import CMASampler
    N, Pk = 37, 4                                              # 37 query rows over 2 ranks: 19 + 18 (ragged)
    bank = torch.zeros(N, 8)
    smp = CMASampler(bank, bank, {"type": "consensus", "pos_k": Pk})
    seen = []

    def fake_range(q0, q1, batch=1024):                         # stands in for the HIP search of rows [q0, q1)
        seen.append((q0, q1))
        rows = torch.arange(q0, q1, dtype=torch.int32).view(-1, 1)
        return rows * 10 + torch.arange(Pk, dtype=torch.int32).view(1, -1)

    smp.sample_range = fake_range
    out = smp.sample()
    return out.numpy(), seen


def test_cma_search_shards_query_rows_and_allgathers_in_rank_order():
    """criterions/avid_cma.py:CMASampler.sample — every rank searches its contiguous shard of query rows, the
    shards are padded to equal length, all-gathered and trimmed: the result is the single-process result, on every
    rank, also when N is not a multiple of the world size."""
    res = run2(_cma_shard_job)
    want = (torch.arange(37, dtype=torch.int32).view(-1, 1) * 10 + torch.arange(4, dtype=torch.int32).view(1, -1)).numpy()
    assert res[0][1] == [(0, 19)] and res[1][1] == [(19, 37)]
    for rank in range(2):
        assert res[rank][0].shape == (37, 4) and (res[rank][0] == want).all()


def _broadcast_job(rank, world):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "avid-cma_amd"))
    from avid_hip.parallel import TrainStep
    torch.manual_seed(1000 + rank)                              # ranks start from DIFFERENT weights / buffers
    m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16), torch.nn.Linear(16, 4))
    m[1].running_mean.fill_(float(rank + 1))
    eng = TrainStep(m, criterion=None)
    views_ok = all(p.data_ptr() == eng.flat.flat.data_ptr() + 4 * o for p, o in zip(eng.flat.params, eng.flat.offsets))
    return eng.flat.flat.clone().numpy(), m[1].running_mean.clone().numpy(), views_ok, eng.buckets.world


def test_trainstep_broadcasts_rank0_parameters_and_buffers():
    """DDP's construction-time broadcast (utils/main_utils.py:112): every rank starts from rank 0's parameters and
    buffers; the parameters are views of the flat buffer afterwards; the 1/world gradient scale is the group size."""
    res = run2(_broadcast_job)
    assert (res[0][0] == res[1][0]).all() and (res[0][1] == res[1][1]).all()
    assert (res[1][1] == 1.0).all()                             # rank 0's buffer value
    torch.manual_seed(1000)
    ref = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16), torch.nn.Linear(16, 4))
    want = torch.cat([p.detach().reshape(-1) for p in reversed(list(ref.parameters()))])
    got = torch.from_numpy(res[1][0])
    assert got.numel() >= want.numel() and res[0][2] and res[1][2] and res[0][3] == 2
    # flat layout: reverse registration order, every slice padded to 4 floats
    off = 0
    for p in reversed(list(ref.parameters())):
        assert torch.equal(got[off:off + p.numel()], p.detach().reshape(-1))
        off += (p.numel() + 3) // 4 * 4


def _buffers_job(rank, world):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "avid-cma_amd"))
    from avid_hip.parallel import TrainStep
    torch.manual_seed(7)
    m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16), torch.nn.Linear(16, 8), torch.nn.BatchNorm1d(8))
    eng = TrainStep(m, criterion=None)
    flat_ok = all(b.data_ptr() == eng.flat_buffers.flat.data_ptr() + 4 * o
                  for b, o in zip(eng.flat_buffers.bufs, eng.flat_buffers.offsets))
    m.train()
    torch.manual_seed(100 + rank)                       # per-rank batches -> per-rank running statistics (no SyncBN)
    for _ in range(3):
        m(torch.randn(32, 8) * (1 + rank) + rank)
    before = [m[1].running_mean.clone(), m[3].running_var.clone(), m[1].num_batches_tracked.clone()]
    eng.sync_buffers()                                  # DDP's buffer broadcast, on request: ONE flat collective
    after = [m[1].running_mean.clone(), m[3].running_var.clone(), m[1].num_batches_tracked.clone()]
    sd_keys = sorted(k for k in m.state_dict() if "running" in k or "num_batches" in k)
    return [t.numpy() for t in before], [t.numpy() for t in after], flat_ok, sd_keys, eng.flat_buffers.numel


def test_sync_buffers_broadcasts_rank0_running_statistics_in_one_flat_buffer():
    """utils/main_utils.py:112 (DistributedDataParallel, broadcast_buffers=True): rank 0's BatchNorm running statistics
    are what every rank evaluates / checkpoints with.  TrainStep keeps them in ONE flat buffer (views; state_dict keys
    unchanged) and broadcasts on request; ranks differ before (per-rank statistics) and equal rank 0 after."""
    res = run2(_buffers_job)
    b0, a0, ok0, keys0, n0 = res[0]
    b1, a1, ok1, keys1, n1 = res[1]
    assert ok0 and ok1 and n0 == n1 == 16 * 2 + 8 * 2
    assert keys0 == ["1.num_batches_tracked", "1.running_mean", "1.running_var", "3.num_batches_tracked", "3.running_mean",
                     "3.running_var"]
    assert not (b0[0] == b1[0]).all() and not (b0[1] == b1[1]).all()      # per-rank statistics before
    for k in range(2):
        assert (a0[k] == b0[k]).all() and (a1[k] == b0[k]).all()          # rank 0's values everywhere after
    assert int(a0[2]) == int(a1[2]) == 3                                  # the step counter advances identically anyway


def test_optimizer_state_dict_indexes_all_parameters_like_torch_adam():
    """utils/main_utils.py:250-261 builds Adam over model.parameters(): state_dict indices count FROZEN parameters too
    (they have an index in param_groups but no state).  TrainStep.state_dict must use the same numbering, and a
    torch.optim.Adam state_dict of the same model must load."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "avid-cma_amd"))
    from avid_hip.parallel import TrainStep
    m = _tiny_model()
    m[2].weight.requires_grad_(False)                            # index 2 of 6 is frozen
    eng = TrainStep(m, criterion=None)
    eng.t = 3
    eng.m.copy_(torch.arange(eng.m.numel(), dtype=torch.float32))
    sd = eng.state_dict()
    assert sd["param_groups"][0]["params"] == [0, 1, 2, 3, 4, 5]
    assert sorted(sd["state"]) == [0, 1, 3, 4, 5]
    params = list(m.parameters())
    for k in sd["state"]:
        assert sd["state"][k]["exp_avg"].shape == params[k].shape
    ref = torch.optim.Adam(m.parameters(), lr=2e-4, weight_decay=1e-5)
    for p in params:
        p.grad = torch.ones_like(p) if p.requires_grad else None
    ref.step()
    rsd = ref.state_dict()
    assert sorted(rsd["state"]) == sorted(sd["state"]) and rsd["param_groups"][0]["params"] == sd["param_groups"][0]["params"]
    eng2 = TrainStep(m, criterion=None)
    eng2.load_state_dict(rsd)
    for k, st in rsd["state"].items():
        i = {id(p): j for j, p in enumerate(eng2.flat.params)}[id(params[k])]
        assert torch.equal(eng2._slice(eng2.m, i), st["exp_avg"]) and torch.equal(eng2._slice(eng2.v, i), st["exp_avg_sq"])
    assert eng2.t == 1


def _wrapper_job(rank, world):
    """The reference's loop (main-avid.py:175-178: zero_grad, backward, step) on two ranks, once with torch's
    DistributedDataParallel and once with avid_hip.parallel.DistributedDataParallel (utils/main_utils.py:112), torch.optim.SGD
    either way; per-rank batches, ranks start from different weights."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "avid-cma_amd"))
    from avid_hip import parallel
    out = {}
    for name, wrap in (("torch", torch.nn.parallel.DistributedDataParallel),
                       ("ours", lambda m: parallel.DistributedDataParallel(m, bucket_cap_mb=256 / (1 << 20)))):
        torch.manual_seed(1000 + rank)
        m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16), torch.nn.ReLU(), torch.nn.Linear(16, 16),
                                torch.nn.ReLU(), torch.nn.Linear(16, 4))
        m[1].running_mean.fill_(float(rank + 1))
        net = wrap(m)
        opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)
        torch.manual_seed(100 + rank)
        seated = []
        for _ in range(3):
            x = torch.randn(6, 8) * (1 + rank)
            loss = net(x).pow(2).sum()
            opt.zero_grad()                               # (set_to_none: `.grad` is None when the backward pass starts)
            loss.backward()
            seated.append(all(p.grad is not None for p in m.parameters()))
            opt.step()
        out[name] = ([p.detach().clone().numpy() for p in m.parameters()], m[1].running_mean.clone().numpy(),
                     sorted(net.state_dict()), seated)
        if name == "ours":
            out["buckets"] = len(net._engine.buckets.bounds)
            out["module_is"] = net.module is m
    return out


def test_dropin_ddp_equals_torch_ddp_on_two_ranks():
    res = run2(_wrapper_job)
    for rank in range(2):
        r = res[rank]
        assert r["buckets"] >= 3 and r["module_is"]
        assert all(r["ours"][3]), "a parameter had no .grad after backward()"
        assert r["ours"][2] == r["torch"][2] and all(k.startswith("module.") for k in r["ours"][2])
        for a, b in zip(r["ours"][0], r["torch"][0]):
            torch.testing.assert_close(torch.from_numpy(a), torch.from_numpy(b), rtol=1e-5, atol=1e-6)
    for a, b in zip(res[0]["ours"][0], res[1]["ours"][0]):
        assert (a == b).all()                               # the ranks stay in lock step
    # buffers: rank 1 follows rank 0's running statistics (broadcast before every training forward), as under torch's
    torch.testing.assert_close(torch.from_numpy(res[1]["ours"][1]), torch.from_numpy(res[1]["torch"][1]), rtol=1e-5, atol=1e-6)


def test_dropin_ddp_refuses_what_it_does_not_reproduce():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "avid-cma_amd"))
    from avid_hip import parallel
    m = _tiny_model()
    with pytest.raises(NotImplementedError):
        parallel.DistributedDataParallel(m, find_unused_parameters=True)
    with pytest.raises(NotImplementedError):
        parallel.DistributedDataParallel(m, static_graph=True)
    net = parallel.DistributedDataParallel(m)
    with pytest.raises(NotImplementedError):
        net.no_sync()
    assert net.module is m and not list(net._engine.buckets.works)
    # eval / no_grad calls go straight to the module
    net.eval()
    with torch.no_grad():
        assert net(torch.zeros(2, 8)).shape == (2, 4)
