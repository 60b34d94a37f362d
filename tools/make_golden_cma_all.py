#!/usr/bin/env python
"""Generate tests/golden/cma_all.npz by IMPORTING the reference (build container only): AVID_CMA with ALL FOUR score
groups active (xModalInst, wModalInst, xModalPos, wModalPos > 0) — the configuration that exercises the
within-modal-instance key quirk of criterions/avid_cma.py:175-177 — on the banks / positive set of cma.npz.

    python tools/make_golden_cma_all.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import os
import sys
import warnings

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import detgen  # noqa: E402


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    torch.Tensor.cuda = lambda s, *a, **k: s
    torch.nn.Module.cuda = lambda s, *a, **k: s
    sys.path.insert(0, args.ref)
    import criterions  # reference
    from criterions.nce import NCECriterion
    from criterions.avid_cma import AVIDSimilarityPositiveExpansion
    torch.manual_seed(0)
    torch.set_num_threads(8)
    g = np.load(os.path.join(args.out, "cma.npz"))

    def det_bank(tag, N, D=128):
        return torch.nn.functional.normalize(T(detgen.det_normalish(f"bank:{tag}", (N, D))), p=2, dim=1)

    N, Pk, K, Kw, bs = 500, 32, 64, 16, 4
    coeffs = (0.4, 0.1, 0.2, 0.3)
    crit = criterions.AVID_CMA.__new__(criterions.AVID_CMA)
    torch.nn.Module.__init__(crit)
    na = AVIDSimilarityPositiveExpansion(memory_size=N, embedding_dim=128, num_negatives=K, num_negatives_within=Kw,
                                         xModalInst=True, wModalInst=True, xModalPos=True, wModalPos=True,
                                         sampling_args={"type": "consensus", "pos_k": Pk}, momentum=0.5)
    na.view1_mem.copy_(det_bank("cma:v1", N)); na.view2_mem.copy_(det_bank("cma:v2", N))
    na.register_buffer("positive_set", T(g["topk_consensus"]).int())
    crit.nce_average = na
    crit.xModalInstCoeff, crit.wModalInstCoeff, crit.xModalPosCoeff, crit.wModalPosCoeff = coeffs
    crit.criterion = NCECriterion(N)
    out = {"coeffs": np.array(coeffs, np.float64)}
    for step in range(2):
        y = T(detgen.det_indices(f"cma_all:y{step}", bs, N))
        rand_idx = T(detgen.det_indices(f"cma_all:draw{step}", bs * K, N - Pk)).view(bs, K)
        na.multinomial.draw = lambda n, _r=rand_idx: _r.reshape(-1)
        v = T(detgen.det_normalish(f"cma_all:v{step}", (bs, 128))).requires_grad_(True)
        a = T(detgen.det_normalish(f"cma_all:a{step}", (bs, 128))).requires_grad_(True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            loss, tb = crit(v, a, y)
        loss.backward()
        out[f"y{step}"], out[f"rand{step}"] = y.numpy(), rand_idx.numpy()
        out[f"loss{step}"] = loss.item()
        out[f"tb_keys{step}"] = np.array(sorted(tb.keys()))
        for k in tb:
            out[f"tb{step}_{k.replace('/', '_')}"] = float(tb[k])
        out[f"gv{step}"], out[f"ga{step}"] = v.grad.numpy().copy(), a.grad.numpy().copy()
        out[f"Z{step}"] = float(crit.criterion.avg_exp_score)
        out[f"v1rows{step}"] = na.view1_mem[y].numpy().copy()
        out[f"v2rows{step}"] = na.view2_mem[y].numpy().copy()
    np.savez_compressed(os.path.join(args.out, "cma_all.npz"), **out)
    print("cma_all.npz", os.path.getsize(os.path.join(args.out, "cma_all.npz")), {k: out[k] for k in out if k.startswith(("loss", "tb_keys"))})


if __name__ == "__main__":
    main()
