"""The full training step with TWO ranks on the one MI355X of the GPU box (BASELINE config 3's semantics).

RCCL refuses two ranks on one device, gloo does not (it stages device tensors through the host), so two
processes share ``cuda:0`` with backend ``gloo`` and each drives ``avid_hip.parallel.TrainStep`` on its own shard.
What the reference does with DistributedDataParallel (utils/main_utils.py:105-117) and the all-gathered bank update
(criterions/avid.py:103-129) is checked against single-process runs of the same kernels and against the oracle:

* construction: every rank starts from rank 0's parameters, buffers and banks (DDP / init_memory broadcasts);
* step 0: Z = mean over ranks of the per-rank mean(exp(neg)) (criterions/nce.py:27-33); each rank's loss is the
  single-process loss of ITS shard with ITS BatchNorm statistics (no SyncBN) and equals the oracle's; the
  all-reduced gradient buffer is bit-identical to g(shard 0) + g(shard 1); the Adam step uses their mean;
  both banks equal ONE update with the records of both ranks in rank order;
* every step: parameters and both banks stay bit-identical across ranks; BatchNorm running statistics do not
  (per-rank statistics) until ``sync_buffers()``; a sample id present on both ranks resolves "highest global
  position wins";
* AVID_CMA: ``find_correspondences`` sharded over the two ranks equals the single-process search bit for bit;
* torch's own DistributedDataParallel around the same modules (what an unmodified main-avid.py does) produces the
  mean of the two shards' gradients.
"""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_BANK, K_NEG, BS = 2000, 64, 2
IDS = [[5, 17, 900, 31], [40, 77, 77, 12], [5, 6, 7, 8]]        # global batches; 77 sits on BOTH ranks in step 1


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _det_bank(tag, N):
    from oracle import detgen
    return torch.nn.functional.normalize(T(detgen.det_normalish(f"bank:{tag}", (N, 128))), p=2, dim=1)


def _inputs():
    from oracle import detgen
    video = T(detgen.det_normalish("two:video", (2 * BS, 3, 8, 64, 64)))
    audio = T(detgen.det_normalish("two:audio", (2 * BS, 1, 40, 100)))
    return video, audio


def _model(dev):
    import models
    from oracle import detgen
    m = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128])
    sd = m.state_dict()
    m.load_state_dict({k: T(detgen.det_param(f"w:{k}", tuple(v.shape)).copy()).to(v.dtype) for k, v in sd.items()})
    return m.to(dev).train()


def _criterion(dev, rank):
    import criterions
    crit = criterions.AVID(num_data=N_BANK, embedding_dim=128, num_negatives=K_NEG, momentum=0.5, device=dev.index)
    return crit


def _set_banks(crit, rank):
    crit.nce_average.view1_mem.copy_(_det_bank("two:v1", N_BANK))
    crit.nce_average.view2_mem.copy_(_det_bank("two:v2", N_BANK))
    crit.nce_average.multinomial.reseed(100 + rank, 0)


class _RecordUpdates:
    """Wraps ops.bank_update / ops.bank_update_pair: keeps (y_all, emb_all, rows before) of every bank's update, in
    the order view1_mem, view2_mem."""

    def __init__(self):
        from avid_hip import ops
        self.ops, self.orig, self.orig_pair, self.calls = ops, ops.bank_update, ops.bank_update_pair, []

    def __enter__(self):
        def wrapped(bank, y, emb, momentum):
            self.calls.append((y.clone().cpu(), emb.clone().cpu(), bank[y].clone().cpu()))
            return self.orig(bank, y, emb, momentum)

        def wrapped_pair(bank0, bank1, y, emb0, emb1, m0, m1):
            self.calls.append((y.clone().cpu(), emb0.clone().cpu(), bank0[y].clone().cpu()))
            self.calls.append((y.clone().cpu(), emb1.clone().cpu(), bank1[y].clone().cpu()))
            return self.orig_pair(bank0, bank1, y, emb0, emb1, m0, m1)
        self.ops.bank_update = wrapped
        self.ops.bank_update_pair = wrapped_pair
        return self

    def __exit__(self, *exc):
        self.ops.bank_update = self.orig
        self.ops.bank_update_pair = self.orig_pair
        return False


# ----------------------------------------------------------------------------------------------- workers
def _train_job(rank, world, out, per_layer=False):
    import torch.distributed as dist
    from avid_hip.parallel import TrainStep
    if per_layer:           # the per-layer autograd path (what a hooked / partly frozen model falls back to): deferred weight
        from avid_hip import plan     # gradients report their buckets with the TRAILING stream current
        plan.ENABLED = False
    dev = torch.device("cuda", 0)
    m = _model(dev)
    if rank == 1:                                    # ranks start DIFFERENT: the construction broadcast must fix it
        with torch.no_grad():
            m.video_model.conv1[0].weight.add_(1.0)
            m.audio_model.conv1[1].running_mean.fill_(3.0)
    crit = _criterion(dev, rank)
    res = {"bank_after_init": crit.nce_average.view1_mem.clone().cpu(),      # randn on each rank, then rank 0's
           "overlap_default": bool(m.overlap_towers)}
    _set_banks(crit, rank)
    eng = TrainStep(m, crit, bucket_bytes=4 << 20, broadcast_buffers="lazy")     # (per-rank statistics stay visible)
    res["overlap_engine"] = bool(m.overlap_towers)
    res["params_after_init"] = eng.flat.flat.clone().cpu()
    res["nbuckets"] = len(eng.buckets.bounds)
    video, audio = _inputs()
    v = video[rank * BS:(rank + 1) * BS].to(dev)
    a = audio[rank * BS:(rank + 1) * BS].to(dev)
    res["loss"], res["bn_mean"], res["banks"], res["updates"] = [], [], [], []
    for step, ids in enumerate(IDS):
        y = torch.tensor(ids[rank * BS:(rank + 1) * BS], dtype=torch.int64, device=dev)
        with _RecordUpdates() as rec:
            loss = eng.forward_backward(v, a, y)
        if step == 0:
            res["grad0"] = eng.flat.grad.clone().cpu()
            res["Z"] = crit.criterion.avg_exp_score.clone().cpu()
        eng.optimizer_step()
        if step == 0:
            res["params0"] = eng.flat.flat.clone().cpu()
        res["loss"].append(float(loss))
        res["bn_mean"].append(m.video_model.conv1[1].running_mean.clone().cpu())
        res["banks"].append((crit.nce_average.view1_mem.clone().cpu(), crit.nce_average.view2_mem.clone().cpu()))
        res["updates"].append(rec.calls)
    res["params_final"] = eng.flat.flat.clone().cpu()
    from avid_hip import streams
    res["stream_sets"] = len(streams._PLACED)
    # an evaluation forward (per-layer path, AV_Wrapper.forward re-seats its buffers) must not orphan the tensor that
    # sync_buffers() broadcasts: ONE owner of the BatchNorm statistics
    m.eval()
    with torch.no_grad():
        m(v, a)
    m.train()
    res["bn_mean_before_sync"] = m.video_model.conv1[1].running_mean.clone().cpu()
    eng.sync_buffers()
    res["bn_mean_synced"] = m.video_model.conv1[1].running_mean.clone().cpu()
    res["bn_all_synced"] = torch.cat([b.detach().flatten().float().cpu() for n, b in m.named_buffers()
                                      if b.is_floating_point() and n != "_bn_flat"])
    res["one_owner"] = all(b.untyped_storage().data_ptr() == eng.flat_buffers.flat.untyped_storage().data_ptr()
                           for n, b in m.named_buffers() if b.is_floating_point())
    res["audio_bn_after_init"] = None
    from avid_hip import ops
    ops.check_device_errors(dev)
    return res


def _cma_job(rank, world, out):
    import criterions
    dev = torch.device("cuda", 0)
    N = 3000
    crit = criterions.AVID_CMA(num_data=N, embedding_dim=128, num_negatives=256, num_negatives_within=64, momentum=0.5,
                               sampling_args={"type": "consensus", "pos_k": 32}, device=0)
    crit.nce_average.view1_mem.copy_(_det_bank("twocma:v1", N))
    crit.nce_average.view2_mem.copy_(_det_bank("twocma:v2", N))
    ptr = crit.nce_average.positive_set.data_ptr()
    crit.nce_average.find_correspondences()          # sharded: rank r searches rows [1500 r, 1500 (r+1))
    res = {"positive_set": crit.nce_average.positive_set.clone().cpu(),
           "in_place": crit.nce_average.positive_set.data_ptr() == ptr}
    crit.nce_average.multinomial.reseed(7 + rank, 0)
    g = torch.Generator().manual_seed(50 + rank)
    v = torch.randn(4, 128, generator=g).to(dev).requires_grad_(True)
    a = torch.randn(4, 128, generator=g).to(dev).requires_grad_(True)
    y = torch.tensor([[3, 1000, 2999, 17], [17, 5, 1500, 8]][rank], device=dev)      # 17 on both ranks
    loss, _ = crit(v, a, y)
    loss.backward()
    res["loss"] = float(loss)
    res["banks"] = (crit.nce_average.view1_mem.clone().cpu(), crit.nce_average.view2_mem.clone().cpu())
    return res


def _ddp_job(rank, world, out):
    """torch's DistributedDataParallel (gloo) around the build's modules — the unmodified main-avid.py path."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    dev = torch.device("cuda", 0)
    m = _model(dev)
    crit = _criterion(dev, rank)
    _set_banks(crit, rank)
    crit.criterion.avg_exp_score.fill_(float(os.environ["TWO_RANK_Z"]))
    from avid_hip import ops
    ops.FUSED_CRITERION = False          # compared bit for bit with _single(), which runs the unfused criterion ops
    if rank == 1:                                    # ranks start with DIFFERENT statistics: DDP's construction-time
        with torch.no_grad():                        # _sync_module_states must install rank 0's (through _bn_flat)
            m.audio_model.conv1[1].running_mean.fill_(3.0)
    ddp = DDP(m, device_ids=[0])
    flat_at_wrap = m._bn_flat.clone().cpu()
    video, audio = _inputs()
    v = video[rank * BS:(rank + 1) * BS].to(dev)
    a = audio[rank * BS:(rank + 1) * BS].to(dev)
    y = torch.tensor(IDS[0][rank * BS:(rank + 1) * BS], dtype=torch.int64, device=dev)
    ve, ae = ddp(v, a)
    loss, _ = crit(ve, ae, y)
    loss.backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.clone().cpu() for n, p in m.named_parameters()}
    # DDP copies rank 0's buffers to every rank before every forward (broadcast_buffers=True, utils/main_utils.py:112): here
    # ONE tensor (models.AV_Wrapper._bn_flat) of which the BatchNorm modules' own buffers are views
    own_stats = {n: b.clone().cpu() for n, b in m.named_buffers() if n.endswith("running_var")}
    n_bcast = sum(1 for n, _ in m.named_buffers() if n not in ddp.parameters_to_ignore)
    # second forward: the broadcast in front of it installs rank 0's statistics on every rank.  The training-mode forward
    # then updates them again from each rank's own shard, so what the broadcast installed is read where it is visible: in
    # a forward pre-hook of the wrapped module (DDP broadcasts before it calls module.forward)
    seen = {}
    def see(mod, args):                              # (a pre-hook that returns something replaces the arguments)
        seen.setdefault("flat", mod._bn_flat.clone().cpu())
    hook = m.register_forward_pre_hook(see)
    with torch.no_grad():
        ddp(v, a)
    hook.remove()
    torch.cuda.synchronize()
    return {"loss": float(loss), "overlap": bool(m.overlap_towers), "grads": grads, "stats_after_step0": own_stats,
            "broadcast_tensors": n_bcast, "flat_after_broadcast": seen["flat"],
            "flat_at_wrap": flat_at_wrap,
            "views": all(b.untyped_storage().data_ptr() == m._bn_flat.untyped_storage().data_ptr()
                         for n, b in m.named_buffers() if b.is_floating_point())}


def _dropin_job(rank, world, out):
    """main-avid.py:169-178 on two ranks with avid_hip.parallel.DistributedDataParallel + avid_hip.parallel.Adam (utils/main_utils.py:112,
    :250), against TrainStep on the same two ranks: the wrapper leaves the MEAN in `.grad` (sum, then / world: exact for two
    ranks) where the engine folds 1 / world into its Adam launch — the same numbers reach the same kernel."""
    from avid_hip import parallel
    dev = torch.device("cuda", 0)
    video, audio = _inputs()
    v = video[rank * BS:(rank + 1) * BS].to(dev)
    a = audio[rank * BS:(rank + 1) * BS].to(dev)
    res = {}
    for which in ("engine", "dropin"):
        m = _model(dev)
        if rank == 1:
            with torch.no_grad():
                m.video_model.conv1[0].weight.add_(1.0)
        crit = _criterion(dev, rank)
        _set_banks(crit, rank)
        losses = []
        if which == "engine":
            eng = parallel.TrainStep(m, crit, bucket_bytes=4 << 20)
            for ids in IDS:
                y = torch.tensor(ids[rank * BS:(rank + 1) * BS], dtype=torch.int64, device=dev)
                losses.append(float(eng.step(v, a, y)))
        else:
            net = parallel.DistributedDataParallel(m, device_ids=[0], bucket_cap_mb=4)
            opt = parallel.Adam(net.parameters(), lr=2e-4, weight_decay=1e-5, betas=[0.9, 0.999])
            res["nbuckets"] = len(net._engine.buckets.bounds)
            res["comm"] = bool(net._engine.buckets.comm)
            for ids in IDS:
                y = torch.tensor(ids[rank * BS:(rank + 1) * BS], dtype=torch.int64, device=dev)
                ve, ae = net(v, a)
                loss, _ = crit(ve, ae, y)
                losses.append(loss.item())
                opt.zero_grad()
                loss.backward()
                opt.step()
            res["program"] = bool([p for p in m.__dict__.get("_avid_plans", {}).values() if p])
        torch.cuda.synchronize()
        res[which] = (losses, {k: t.clone().cpu() for k, t in m.state_dict().items()},
                      (crit.nce_average.view1_mem.clone().cpu(), crit.nce_average.view2_mem.clone().cpu()))
    return res


def _train_job_per_layer(rank, world, out):
    return _train_job(rank, world, out, per_layer=True)


_JOBS = {"train": _train_job, "train_pl": _train_job_per_layer, "cma": _cma_job, "ddp": _ddp_job, "dropin": _dropin_job}


def _worker(rank, world, port, out, job):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = _JOBS[job](rank, world, out)
        torch.save(res, os.path.join(out, f"{job}_{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _run2(job, out, timeout=900):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(out), job)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(timeout) for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return [torch.load(os.path.join(str(out), f"{job}_{r}.pt"), weights_only=False) for r in range(2)]


# ----------------------------------------------------------------------------------------------- single-process side
def _single(gpu_device, rank, Z):
    """The same kernels on ONE rank's shard of step 0 (no process group): loss, gradient buffer, Z of the shard,
    normalised embeddings handed to the bank update."""
    from avid_hip.parallel import TrainStep
    m = _model(gpu_device)
    crit = _criterion(gpu_device, rank)
    _set_banks(crit, rank)
    if Z is not None:
        crit.criterion.avg_exp_score.fill_(Z)
    eng = TrainStep(m, crit, bucket_bytes=4 << 20)
    video, audio = _inputs()
    v = video[rank * BS:(rank + 1) * BS].to(gpu_device)
    a = audio[rank * BS:(rank + 1) * BS].to(gpu_device)
    y = torch.tensor(IDS[0][rank * BS:(rank + 1) * BS], dtype=torch.int64, device=gpu_device)
    # (the two-rank job's step 0 estimates Z and therefore runs the unfused criterion ops; with Z preset this run would
    # take the fused kernel, whose loss differs in the last bits — compare like with like)
    from avid_hip import ops
    fused, ops.FUSED_CRITERION = ops.FUSED_CRITERION, False
    try:
        with _RecordUpdates() as rec:
            loss = eng.forward_backward(v, a, y)
    finally:
        ops.FUSED_CRITERION = fused
    return {"loss": float(loss), "grad": eng.flat.grad.clone(), "Z": crit.criterion.avg_exp_score.clone().cpu(),
            "emb": [c[1] for c in rec.calls], "eng": eng, "model": m}


def test_two_rank_training_step(tmp_path, gpu_device):
    from oracle import avid_oracle as O
    from avid_hip import ops
    r = _run2("train", tmp_path)
    # ---- construction: rank 0's state everywhere
    assert r[0]["overlap_default"] is False and r[0]["overlap_engine"] is True
    assert torch.equal(r[0]["bank_after_init"], r[1]["bank_after_init"])
    assert torch.equal(r[0]["params_after_init"], r[1]["params_after_init"])
    assert r[0]["nbuckets"] >= 4
    # ---- Z: mean over ranks of the per-rank first-batch means (criterions/nce.py:27-33)
    s0, s1 = _single(gpu_device, 0, None), _single(gpu_device, 1, None)
    Z = (s0["Z"] + s1["Z"]) / 2
    assert torch.equal(r[0]["Z"], Z) and torch.equal(r[1]["Z"], Z)
    assert float(s0["Z"]) != float(s1["Z"])
    # ---- step 0 with the shared Z: per-rank loss / gradient are the single-process ones of that shard
    s0, s1 = _single(gpu_device, 0, float(Z)), _single(gpu_device, 1, float(Z))
    assert r[0]["loss"][0] == s0["loss"] and r[1]["loss"][0] == s1["loss"]
    gsum = (s0["grad"] + s1["grad"]).cpu()
    assert torch.equal(r[0]["grad0"], gsum) and torch.equal(r[1]["grad0"], gsum)      # every bucket, bit for bit
    # Adam on the MEAN gradient (1/world folded into the kernel)
    e = s0["eng"]
    e.flat.grad.copy_(gsum.to(gpu_device))
    ops.adam_flat(e.flat.flat, e.flat.grad, e.m, e.v, e.lr, e.betas[0], e.betas[1], e.eps, e.wd, 1, grad_scale=0.5)
    assert torch.equal(r[0]["params0"], e.flat.flat.cpu()) and torch.equal(r[1]["params0"], r[0]["params0"])
    # ---- the oracle on each shard separately, per-shard BatchNorm statistics, shared Z
    video, audio = _inputs()
    P = O.det_state(O.av_wrapper_spec(18), "w")
    prob, alias = O.alias_build_uniform(N_BANK - 1)
    for rank in range(2):
        y = torch.tensor(IDS[0][rank * BS:(rank + 1) * BS])
        idx = T(O.sample_negatives_from_draw(O.alias_draw_philox(prob, alias, BS * K_NEG, 100 + rank, 0), y.numpy(), K_NEG))
        with torch.no_grad():
            ve, ae = O.av_forward(video[rank * BS:(rank + 1) * BS], audio[rank * BS:(rank + 1) * BS],
                                  {k: v.clone() for k, v in P.items()}, 18, True)
            ref, _, _ = O.avid_forward(ve, ae, y, idx, _det_bank("two:v1", N_BANK), _det_bank("two:v2", N_BANK),
                                       float(Z), 0.5)
        np.testing.assert_allclose(r[rank]["loss"][0], float(ref), rtol=2e-5)
    # ---- banks after step 0 == ONE update with both ranks' records in rank order
    for b, tag in enumerate(["two:v1", "two:v2"]):
        bank = _det_bank(tag, N_BANK).to(gpu_device)
        ops.bank_update(bank, torch.tensor(IDS[0], device=gpu_device),
                        torch.cat([s0["emb"][b], s1["emb"][b]]).to(gpu_device), 0.5)
        assert torch.equal(r[0]["banks"][0][b], bank.cpu())
    # ---- every step: replicas bit-identical, BatchNorm statistics per rank
    for step in range(len(IDS)):
        for b in range(2):
            assert torch.equal(r[0]["banks"][step][b], r[1]["banks"][step][b])
        assert not torch.equal(r[0]["bn_mean"][step], r[1]["bn_mean"][step])          # no SyncBN
        assert np.isfinite(r[0]["loss"][step]) and np.isfinite(r[1]["loss"][step])
    assert torch.equal(r[0]["params_final"], r[1]["params_final"])
    assert not torch.equal(r[0]["params_final"], r[0]["params0"])
    assert torch.equal(r[0]["bn_mean_synced"], r[0]["bn_mean"][-1])                   # rank 0's statistics win
    assert torch.equal(r[1]["bn_mean_synced"], r[0]["bn_mean"][-1])
    # ... although an evaluation forward ran on both ranks before sync_buffers(): the buffers and the tensor that is
    # broadcast have ONE owner (every statistic of rank 1 equals rank 0's afterwards, not only the probed one)
    assert not torch.equal(r[1]["bn_mean_before_sync"], r[0]["bn_mean_before_sync"])
    assert torch.equal(r[0]["bn_all_synced"], r[1]["bn_all_synced"]) and r[0]["one_owner"] and r[1]["one_owner"]
    assert r[0]["stream_sets"] == 1 and r[1]["stream_sets"] == 1
    # ---- the same job on the per-layer path (deferred weight gradients report their buckets from the trailing stream):
    #      one StreamSet, every bucket on its one comm stream, the same all-reduced gradients and parameters bit for bit
    rp = _run2("train_pl", tmp_path)
    for rank in range(2):
        assert rp[rank]["stream_sets"] == 1
        assert torch.equal(rp[rank]["grad0"], r[rank]["grad0"])
        assert torch.equal(rp[rank]["params_final"], r[rank]["params_final"])
        assert rp[rank]["loss"] == r[rank]["loss"]
    # ---- all-gathered records arrive in rank order on both ranks; duplicate id 77 (global positions 1 and 2):
    #      the highest global position — rank 1's sample — owns the row (criterions/avid.py:108-129)
    for rank in range(2):
        y_all, emb_all, before = r[rank]["updates"][1][0]
        assert y_all.tolist() == IDS[1]
        assert torch.equal(emb_all, r[1 - rank]["updates"][1][0][1])
        want = torch.nn.functional.normalize(0.5 * before[2] + 0.5 * emb_all[2], dim=0)
        lose = torch.nn.functional.normalize(0.5 * before[1] + 0.5 * emb_all[1], dim=0)
        got = r[rank]["banks"][1][0][77]
        assert float((got - want).abs().max()) < 1e-6 and float((got - lose).abs().max()) > 1e-3
    # ---- torch's DistributedDataParallel around the same modules: mean of the two shards' gradients
    os.environ["TWO_RANK_Z"] = repr(float(Z))
    d = _run2("ddp", tmp_path)
    assert d[0]["overlap"] is False
    assert d[0]["loss"] == s0["loss"] and d[1]["loss"] == s1["loss"]
    # DDP's per-forward buffer broadcast is ONE tensor (the flat BatchNorm statistics); the modules' buffers are views of it;
    # before any broadcast of real values the ranks' statistics differ (per-rank batches)
    assert d[0]["broadcast_tensors"] == 1 and d[0]["views"] and d[1]["views"]
    k0 = next(iter(d[0]["stats_after_step0"]))
    assert not torch.equal(d[0]["stats_after_step0"][k0], d[1]["stats_after_step0"][k0])
    # ... the wrap installed rank 0's statistics on rank 1 (which had been set to something else), and the broadcast in
    # front of the second forward installed rank 0's step-0 statistics on rank 1
    assert torch.equal(d[0]["flat_at_wrap"], d[1]["flat_at_wrap"]) and float(d[1]["flat_at_wrap"].abs().max()) > 0
    assert float(d[1]["flat_at_wrap"].max()) < 3.0
    assert torch.equal(d[0]["flat_after_broadcast"], d[1]["flat_after_broadcast"])
    assert not torch.equal(d[0]["flat_after_broadcast"], d[0]["flat_at_wrap"])
    eng = s1["eng"]
    for i, p in enumerate(eng.flat.params):
        name = next(n for n, q in s1["model"].named_parameters() if q is p)
        o = eng.flat.offsets[i]
        want = (gsum[o:o + p.numel()] / 2).as_strided(p.shape, p.stride())
        for rank in range(2):
            got = d[rank]["grads"][name]
            assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max()) + 1e-12, name


def test_two_rank_cma_search_and_step(tmp_path, gpu_device):
    """find_correspondences sharded 2-way == the single-process search, bit for bit, on both ranks; positive_set is
    refreshed in place (a captured graph keeps its pointer); after a CMA step the bank replicas are identical."""
    from avid_hip import topk
    r = _run2("cma", tmp_path)
    N = 3000
    full = topk.cma_topk(_det_bank("twocma:v1", N).to(gpu_device), _det_bank("twocma:v2", N).to(gpu_device), 0, N, 32,
                         0, batch=1024).cpu()
    for rank in range(2):
        assert r[rank]["in_place"]
        assert torch.equal(r[rank]["positive_set"], full.int())
        assert np.isfinite(r[rank]["loss"])
    for b in range(2):
        assert torch.equal(r[0]["banks"][b], r[1]["banks"][b])
    assert not torch.equal(r[0]["banks"][0][17], _det_bank("twocma:v1", N)[17])


def test_two_rank_reference_loop_with_the_dropin_objects(tmp_path, gpu_device):
    """The reference's loop with its two factory lines swapped, on two ranks: losses, parameters, BatchNorm buffers and banks
    of every rank equal TrainStep's on the same ranks bit for bit; the ranks' parameters stay equal to each other."""
    r = _run2("dropin", tmp_path)
    for rank in range(2):
        assert r[rank]["comm"] and r[rank]["nbuckets"] >= 3 and r[rank]["program"]
        le, sde, be = r[rank]["engine"]
        ld, sdd, bd = r[rank]["dropin"]
        assert ld == le
        for k in sde:
            assert torch.equal(sde[k], sdd[k]), (rank, k)
        assert torch.equal(be[0], bd[0]) and torch.equal(be[1], bd[1])
    names = [k for k, _ in r[0]["dropin"][1].items() if "running" not in k and "num_batches" not in k and k != "_bn_flat"]
    for k in names:
        assert torch.equal(r[0]["dropin"][1][k], r[1]["dropin"][1][k]), k
