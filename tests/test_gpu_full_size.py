"""BASELINE configs 4 and 5 at FULL size on the GPU.

* config 5 — ``configs/main/avid/audioset/Cross-N1024.yaml`` (num_data 1 784 108 -> a 2M x 128 bank, 1.02 GB per
  modality, far beyond the 256 MB of L2/MALL): ``bank_scores`` fwd / bwd and ``bank_update`` against the oracle's
  gather + bmm (criterions/avid.py:56-66,118-129) at bs = 64, K = 1024, plus one whole AVID criterion step with the
  size-independent properties (determinism, only the batch's rows move, rows stay unit vectors, the sampler covers
  the whole range).
* config 4 — ``configs/main/avid-cma/kinetics/InstX-N1024-PosW-N64-Top32.yaml:47-62`` (240k rows, consensus top-32,
  1024 negatives, 64 within-modal negatives): the correspondence search for 256 sampled queries against the oracle's
  dense search, with the threshold filter's overflow counter asserted clear, and one AVID_CMA step at bs = 64.
"""
import numpy as np
import pytest
import torch

from oracle import avid_oracle as O

pytestmark = pytest.mark.gpu

T_INV = 1.0 / 0.07


def _gpu_bank(N, seed, dev):
    g = torch.Generator(device=dev).manual_seed(seed)
    return torch.nn.functional.normalize(torch.randn(N, 128, generator=g, device=dev), p=2, dim=1)


def test_bank_scores_and_update_at_2m_rows(gpu_device):
    """Tolerance: 2e-5 of the score scale (|s| <= 1/T = 14.3) against a float64 gather + bmm; the update to
    1e-6 absolute on unit rows; rows outside the batch bit-identical."""
    from avid_hip import ops
    N, bs, K, D = 2_000_000, 64, 1024, 128
    bank = _gpu_bank(N, 1, gpu_device)
    g = torch.Generator().manual_seed(2)
    emb = torch.nn.functional.normalize(torch.randn(bs, D, generator=g), dim=1)
    y = torch.randperm(N, generator=g)[:bs]
    idx = torch.randint(0, N, (bs, K), generator=g)
    idx[0, 0], idx[0, 1], idx[63, 1023] = 0, N - 1, N - 1          # both ends of the table
    rows = torch.cat([y[:, None], idx], 1)
    ed = emb.to(gpu_device).requires_grad_(True)
    s = ops.bank_scores(ed, bank, rows.to(gpu_device), T_INV)
    ds = torch.randn(bs, K + 1, generator=g)
    s.backward(ds.to(gpu_device))
    gathered = bank[rows.to(gpu_device)].cpu().double()             # [bs, K+1, D] (test plumbing)
    ref = torch.einsum("brd,bd->br", gathered, emb.double()) * T_INV
    assert float((s.detach().cpu().double() - ref).abs().max()) < 2e-5 * T_INV
    gref = torch.einsum("br,brd->bd", ds.double(), gathered) * T_INV
    assert float((ed.grad.cpu().double() - gref).abs().max()) < 2e-5 * float(gref.abs().max())
    ops.check_device_errors(gpu_device)
    # EMA update incl. a duplicate id (the later sample owns the row)
    y2 = y.clone()
    y2[3] = y2[1]
    before = bank.clone()
    ops.bank_update(bank, y2.to(gpu_device), ed.detach(), 0.5)
    v1 = before[y2.to(gpu_device)].cpu()
    want = torch.nn.functional.normalize(0.5 * v1 + 0.5 * emb, p=2, dim=1)
    got = bank[y2.to(gpu_device)].cpu()
    keep = torch.ones(bs, dtype=torch.bool)
    keep[1] = False                                                  # position 1 lost to position 3
    assert float((got[keep] - want[keep]).abs().max()) < 1e-6
    assert float((got[1] - want[3]).abs().max()) < 1e-6
    untouched = torch.ones(N, dtype=torch.bool, device=gpu_device)
    untouched[y2.to(gpu_device)] = False
    assert torch.equal(bank[untouched], before[untouched])


def test_avid_step_at_2m_rows(gpu_device):
    import criterions
    N, bs, K = 2_000_000, 64, 1024

    def one():
        crit = criterions.AVID(num_data=N, embedding_dim=128, num_negatives=K, momentum=0.5, device=gpu_device.index)
        crit.nce_average.view1_mem.copy_(_gpu_bank(N, 11, gpu_device))
        crit.nce_average.view2_mem.copy_(_gpu_bank(N, 12, gpu_device))
        crit.nce_average.multinomial.reseed(42, 0)
        g = torch.Generator().manual_seed(3)
        v = torch.randn(bs, 128, generator=g).to(gpu_device).requires_grad_(True)
        a = torch.randn(bs, 128, generator=g).to(gpu_device).requires_grad_(True)
        y = torch.randperm(N, generator=g)[:bs].to(gpu_device)
        idx = crit.nce_average.sample_negatives(y, K)
        crit.nce_average.multinomial.reseed(42, 0)
        before = crit.nce_average.view2_mem.clone()
        loss, tb = crit(v, a, y)
        loss.backward()
        return crit, float(loss), v.grad.clone(), a.grad.clone(), y, idx, before

    c1, l1, gv1, ga1, y, idx, before = one()
    c2, l2, gv2, ga2, _, _, _ = one()
    assert np.isfinite(l1) and l1 == l2 and torch.equal(gv1, gv2) and torch.equal(ga1, ga2)      # no atomics anywhere
    assert torch.equal(c1.nce_average.view1_mem, c2.nce_average.view1_mem)
    # the sampler covers the whole 2M range and never returns the sample itself
    assert int(idx.min()) >= 0 and int(idx.max()) < N and int(idx.max()) > N - N // 500 and int(idx.min()) < N // 500
    assert not bool((idx == y[:, None]).any())
    # only the batch's rows moved, and they are unit vectors
    changed = (c1.nce_average.view2_mem != before).any(1).nonzero().flatten()
    assert set(changed.tolist()) == set(y.tolist())
    np.testing.assert_allclose(c1.nce_average.view2_mem[y].norm(dim=1).cpu().numpy(), 1.0, rtol=1e-5)
    # the loss against the oracle on the same negatives (pre-update rows, first-call Z)
    v1, v2 = _gpu_bank(N, 11, gpu_device), _gpu_bank(N, 12, gpu_device)
    g = torch.Generator().manual_seed(3)
    v = torch.randn(bs, 128, generator=g)
    a = torch.randn(bs, 128, generator=g)
    rows = torch.cat([y[:, None], idx], 1)
    # compact banks: only the gathered rows, re-indexed (the oracle runs on the CPU)
    uniq, inv = torch.unique(rows, return_inverse=True)
    ref, _, Z = O.avid_forward(v, a, inv[:, 0].cpu(), inv[:, 1:].cpu(), v1[uniq].cpu(), v2[uniq].cpu(), None, 0.5)
    np.testing.assert_allclose(l1, float(ref), rtol=1e-5)
    np.testing.assert_allclose(float(c1.criterion.avg_exp_score), float(Z), rtol=1e-5)


def test_cma_search_at_240k_rows(gpu_device):
    """256 sampled queries (four ranges incl. both ends of the bank) vs the oracle's dense search: rows sorted, self
    excluded, set equality >= 0.998 of the queries (fp32 summation order may swap a boundary pair), every row within
    one element of the oracle's; the filter's candidate lists never overflow on random unit vectors."""
    from avid_hip import topk
    N, Pk = 240_000, 32
    v1, v2 = _gpu_bank(N, 21, gpu_device), _gpu_bank(N, 22, gpu_device)
    c1, c2 = v1.cpu(), v2.cpu()
    fb = torch.zeros((), dtype=torch.int32, device=gpu_device)
    same, total = 0, 0
    for q0 in (0, 77_777, 200_000, N - 64):
        got = topk.cma_topk(v1, v2, q0, q0 + 64, Pk, 0, batch=1024, fallbacks=fb).cpu().numpy()
        want = O.cma_topk(c1, c2, Pk, "consensus", chunk=64, queries=(q0, q0 + 64))[q0:q0 + 64]
        assert (np.diff(got, axis=1) > 0).all()
        for i in range(64):
            sg, sw = set(got[i].tolist()), set(want[i].tolist())
            assert q0 + i not in sg and len(sg & sw) >= Pk - 1
            same += sg == sw
            total += 1
    assert same / total >= 0.998, same / total
    assert int(fb) == 0                                       # the threshold filter handled every batch


def test_avid_cma_step_at_240k_rows(gpu_device):
    """One InstX-N1024-PosW-N64-Top32 step at bs = 64 on a 240k bank (positive_set injected: the search itself is the
    test above): loss and d loss / d emb against the oracle on the same random draws."""
    import criterions
    from criterions.avid_cma import AVIDSimilarityPositiveExpansion
    from criterions.nce import NCECriterion
    N, Pk, K, Kw, bs = 240_000, 32, 1024, 64, 64
    crit = criterions.AVID_CMA.__new__(criterions.AVID_CMA)
    torch.nn.Module.__init__(crit)
    na = AVIDSimilarityPositiveExpansion(memory_size=N, embedding_dim=128, num_negatives=K, num_negatives_within=Kw,
                                         sampling_args={"type": "consensus", "pos_k": Pk}, momentum=0.5,
                                         device=gpu_device.index)
    v1, v2 = _gpu_bank(N, 31, gpu_device), _gpu_bank(N, 32, gpu_device)
    na.view1_mem.copy_(v1)
    na.view2_mem.copy_(v2)
    g = torch.Generator().manual_seed(5)
    pset = torch.stack([torch.randperm(N, generator=g)[:Pk].sort().values for _ in range(bs)])
    y = torch.randperm(N, generator=g)[:bs]
    full = torch.zeros((N, Pk), dtype=torch.int32)
    full[y] = pset.int()
    na.register_buffer("positive_set", full.to(gpu_device))
    crit.nce_average = na
    crit.xModalInstCoeff, crit.wModalInstCoeff, crit.xModalPosCoeff, crit.wModalPosCoeff = 0.5, 0.0, 0.0, 0.5
    crit.criterion = NCECriterion(N).to(gpu_device)
    rand_idx = torch.randint(0, N - Pk, (bs, K), generator=g)
    na.multinomial.draw = lambda n, _r=rand_idx.to(gpu_device): _r.reshape(-1)
    v = torch.randn(bs, 128, generator=g)
    a = torch.randn(bs, 128, generator=g)
    vd, ad = v.to(gpu_device).requires_grad_(True), a.to(gpu_device).requires_grad_(True)
    loss, tb = crit(vd, ad, y.to(gpu_device))
    loss.backward()
    assert set(tb) == {"Loss/inst-v2a", "Loss/inst-a2v", "Loss/pos-v2v", "Loss/pos-a2a"}
    # oracle on compacted banks (only the rows this step touches)
    pos_idx, neg_idx = O.cma_memory_sampling(full, y, rand_idx)
    rows = torch.cat([y[:, None], pos_idx, neg_idx], 1)
    uniq, inv = torch.unique(rows, return_inverse=True)
    inv = inv.cpu()
    vr, ar = v.clone().requires_grad_(True), a.clone().requires_grad_(True)
    sc, v_hat, a_hat = O.cma_scores(vr, ar, inv[:, 0], inv[:, 1:1 + Pk], inv[:, 1 + Pk:], v1[uniq.to(gpu_device)].cpu(),
                                    v2[uniq.to(gpu_device)].cpu(), num_negatives_within=Kw)
    Z, total = None, 0.0
    for k in sc:
        l, Z = O.nce_loss(sc[k][0], sc[k][1], Z)
        total = total + l / 2.0 * 0.5
        np.testing.assert_allclose(float(tb[f"Loss/{k}"]), float(l), rtol=2e-5)
    total.backward()
    np.testing.assert_allclose(float(loss), float(total), rtol=2e-5)
    assert float((vd.grad.cpu() - vr.grad).abs().max()) < 5e-4 * float(vr.grad.abs().max())
    assert float((ad.grad.cpu() - ar.grad).abs().max()) < 5e-4 * float(ar.grad.abs().max())
