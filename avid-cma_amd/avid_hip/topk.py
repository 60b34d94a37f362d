"""CMA correspondence search driver (criterions/avid_cma.py:42-123) on the avid_cma_topk kernels."""
import ctypes as C

import torch

from . import lib
from .ops import _p, _stream, workspace, AvidHipError


def cma_topk(view1, view2, q0, q1, pos_k, kind, batch=128, fallbacks=None):
    """positive_set rows for queries [q0, q1): int32 [q1-q0, pos_k], each row sorted ascending.

    ``kind``: 0 consensus (min), 1 union (max), 2 video, 3 audio.  Queries are processed ``batch`` at a
    time (a multiple of 64) so the [N, batch] score slab stays cache-resident; the last batch is shifted
    back to end at N (a few queries are recomputed) because the kernel wants full batches.
    ``fallbacks`` (optional 0-d int32 device tensor): counts the batches that overflowed the threshold filter
    and were redone by the exact scan (diagnostics; the result is exact either way).
    """
    if not (view1.is_cuda and view2.is_cuda):
        raise AvidHipError("cma_topk needs HIP device tensors")
    N, D = view1.shape
    batch = max(64, (min(batch, N) // 64) * 64)
    if N < 64:
        raise AvidHipError("cma_topk: the bank needs at least 64 rows")
    out = torch.empty((q1 - q0, pos_k), dtype=torch.int32, device=view1.device)
    if q1 <= q0:
        return out
    nb = lib.raw("avid_cma_topk_workspace_bytes")(N, batch, pos_k)
    ws = workspace(view1.device, nb)
    tmp = torch.empty((batch, pos_k), dtype=torch.int32, device=view1.device)
    st = _stream()
    v1, v2 = view1.contiguous(), view2.contiguous()
    q = q0
    while q < q1:
        start = min(q, N - batch)                      # keep a full batch inside the bank
        lib.call("avid_cma_topk", N, D, _p(v1), _p(v2), start, batch, pos_k, int(kind), _p(tmp), _p(fallbacks), _p(ws),
                 ws.numel(), st)
        lo = q - start
        n = min(batch - lo, q1 - q)
        out[q - q0:q - q0 + n] = tmp[lo:lo + n]
        q += n
    return out
