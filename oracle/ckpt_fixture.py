"""Test infrastructure: the regeneration rule of tests/golden/checkpoint.npz (tools/make_golden_checkpoint.py lets the
REFERENCE's CheckpointManager write a checkpoint whose every tensor holds the name-keyed value below; the fixture stores
key order, shapes, dtypes and one sha256 per tensor; tests/test_gpu_checkpoint.py rebuilds the tensors from here)."""
import hashlib

import numpy as np
import torch

from oracle import detgen


def det_tensor(name, like):
    """Deterministic content for checkpoint entry ``name`` (shared with tests/test_gpu_checkpoint.py)."""
    shape = tuple(like.shape)
    if like.dtype == torch.int64:
        return torch.full(shape, 5, dtype=torch.int64)
    if name.endswith("exp_avg_sq") or name.endswith("running_var"):
        v = np.abs(detgen.det_uniform("ckp:" + name, shape)) * 1e-3 + 1e-6
    elif name.endswith(".step") or name.endswith("avg_exp_score"):
        v = np.full(shape, 7.0 if name.endswith(".step") else 1234.5, dtype=np.float32)
    elif "view1_mem" in name or "view2_mem" in name:
        v = detgen.det_normalish("ckp:" + name, shape)
        v = v / np.linalg.norm(v, axis=1, keepdims=True)
    else:
        v = detgen.det_uniform("ckp:" + name, shape) * 0.05
    return torch.from_numpy(np.ascontiguousarray(v.astype(np.float32))).to(like.dtype).reshape(shape)


def sha(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()


def flatten(ckp):
    """[(path, tensor-or-scalar)] in the checkpoint's own order."""
    out = []
    for k in ("model", "train_criterion"):
        for n, v in ckp[k].items():
            out.append((f"{k}/{n}", v))
    for idx, st in ckp["optimizer"]["state"].items():
        for n, v in st.items():
            out.append((f"optimizer/state/{idx}.{n}", v))
    return out


