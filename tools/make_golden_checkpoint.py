#!/usr/bin/env python
"""tests/golden/checkpoint.npz: the STRUCTURE and content hashes of a checkpoint written by the reference's own
``CheckpointManager.save`` (utils/main_utils.py:265-323) for the reference ``av_wrapper`` (under the ``module.`` wrapper
of utils/main_utils.py:105-117), the reference ``AVID`` criterion (criterions/avid.py:146-236, 64-row banks) and the
``torch.optim.Adam`` that ``build_optimizer`` makes (utils/main_utils.py:250-261) after one training step.

Runs only in the build container (imports /root/reference; ``.cuda()`` neutralised in this process as in make_golden.py).
Every tensor of the checkpoint is overwritten with a name-keyed deterministic value (oracle/detgen.py) before the reference
writes it, so the fixture holds no weights — only key order, shapes, dtypes, scalars and a sha256 per tensor; the test
regenerates the values, checks them against the hashes (i.e. against what the reference wrote), loads them through the
build's ``load_state_dict`` paths and compares what the build saves."""
import json
import os
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import detgen  # noqa: E402

REF = "/root/reference"


from oracle.ckpt_fixture import det_tensor, sha, flatten  # noqa: E402


def main():
    torch.Tensor.cuda = lambda s, *a, **k: s
    torch.nn.Module.cuda = lambda s, *a, **k: s
    sys.path.insert(0, REF)
    import models                      # reference
    import criterions                  # reference
    from utils import main_utils       # reference
    torch.manual_seed(0)
    torch.set_num_threads(8)
    N = 64
    model = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128])
    model = torch.nn.DataParallel(model)          # the 'module.' prefix of utils/main_utils.py:105-117 (no GPUs: pass-through)
    crit = criterions.AVID(num_data=N, embedding_dim=128, num_negatives=16, momentum=0.5, xModal_coeff=1., wModal_coeff=0.,
                           device=0)
    params = list(model.parameters()) + list(crit.parameters())      # main-avid.py:105-108
    opt = torch.optim.Adam(params, lr=2e-4, betas=(0.9, 0.999), weight_decay=1e-5)   # utils/main_utils.py:250-256
    g = torch.Generator().manual_seed(1)
    v = torch.randn(2, 3, 8, 32, 32, generator=g)
    a = torch.randn(2, 1, 40, 100, generator=g)
    ve, ae = model(v, a)
    loss, _ = crit(ve, ae, torch.tensor([3, 9]))
    opt.zero_grad()
    loss.backward()
    opt.step()                                                       # Adam state exists now
    # every tensor of what will be saved gets its name-keyed value
    with torch.no_grad():
        for n, t in model.state_dict().items():
            t.copy_(det_tensor("model/" + n, t))
        for n, t in crit.state_dict().items():
            t.copy_(det_tensor("train_criterion/" + n, t))
        for idx, st in opt.state.items():
            pass
        for idx, st in opt.state_dict()["state"].items():
            for n, t in st.items():
                t.copy_(det_tensor(f"optimizer/state/{idx}.{n}", t))
    with tempfile.TemporaryDirectory() as d:
        main_utils.CheckpointManager(d, rank=0).save(3, model=model, optimizer=opt, train_criterion=crit)
        ckp = torch.load(os.path.join(d, "checkpoint.pth.tar"), map_location="cpu", weights_only=False)
    assert list(ckp) == ["epoch", "model", "optimizer", "train_criterion"], list(ckp)
    entries = flatten(ckp)
    meta = {"top_keys": list(ckp), "epoch": ckp["epoch"],
            "param_groups": ckp["optimizer"]["param_groups"],
            "state_indices": list(ckp["optimizer"]["state"].keys()),
            "entries": [[p, list(t.shape), str(t.dtype).replace("torch.", ""), sha(t)] for p, t in entries]}
    # the regeneration rule reproduces what the reference wrote
    for p, t in entries:
        assert torch.equal(det_tensor(p, t), t), p
    out = os.path.join(REPO, "tests", "golden", "checkpoint.npz")
    np.savez_compressed(out, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    print(out, os.path.getsize(out), "bytes;", len(entries), "tensors; param_groups", meta["param_groups"][0].keys())


if __name__ == "__main__":
    main()
