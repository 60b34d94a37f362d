"""Walker/Vose alias sampler (reference: utils/alias_method.py:12-71).

Table construction follows the reference's pairing order exactly (so ``prob`` / ``alias`` are
bit-identical), with a closed form for the all-equal table the criterion actually uses
(``AliasMethod(torch.ones(N-1))``, criterions/avid.py:38 — an O(N) *interpreter* loop in the
reference: ~1 min at N = 2M).  ``draw`` runs the ``avid_alias_draw`` HIP kernel (Philox4x32-10
counter RNG; torch's RNG stream is not reproduced — see DESIGN.md).
"""
import numpy as np
import torch

_DRAW_EPOCH = [0]


class AliasMethod(object):
    def __init__(self, probs):
        if probs.sum() > 1:
            probs.div_(probs.sum())
        K = len(probs)
        p = probs.detach().cpu().numpy().astype(np.float32)
        self.uniform = bool(K > 0 and (p == p[0]).all())
        if self.uniform:
            # all K*p equal => one of smaller/larger is empty, no pairing happens, leftovers -> 1 (alias_method.py:49-50)
            self.prob = torch.ones(K)
            self.alias = torch.zeros(K, dtype=torch.long)
        else:
            prob = np.zeros(K, dtype=np.float32)
            alias = np.zeros(K, dtype=np.int64)
            smaller, larger = [], []
            for kk in range(K):
                prob[kk] = np.float32(K) * p[kk]
                (smaller if prob[kk] < 1.0 else larger).append(kk)
            while len(smaller) > 0 and len(larger) > 0:
                small, large = smaller.pop(), larger.pop()
                alias[small] = large
                prob[large] = np.float32(np.float32(prob[large] - np.float32(1.0)) + prob[small])
                (smaller if prob[large] < 1.0 else larger).append(large)
            for last_one in smaller + larger:
                prob[last_one] = 1
            self.prob = torch.from_numpy(prob)
            self.alias = torch.from_numpy(alias)
        self.seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        self.offset = 0
        self.offset_dev = None      # device-resident draw counter (created by .to(cuda)); hipGraph-replay safe

    def to(self, device):
        self.prob = self.prob.to(device)
        self.alias = self.alias.to(device)
        if self.prob.is_cuda:
            self.offset_dev = torch.full((), self.offset, dtype=torch.int64, device=self.prob.device)

    def cuda(self, device=None):          # reference-era call sites use .cuda()
        self.to(torch.device("cuda", torch.cuda.current_device() if device is None else device))

    def draw(self, N, y=None, per_row=1):
        """Draw N samples (int64, on ``self.prob.device``).

        ``y`` (optional, [N / per_row] int64) fuses the criterion's "avoid self" shift
        ``idx += (idx >= y)`` (criterions/avid.py:85) into the same kernel.
        """
        from avid_hip import ops
        K = self.alias.size(0)
        out = ops.alias_draw(int(N), K, self.prob, self.alias, self.uniform, self.seed, self.offset, y, per_row,
                             device=self.prob.device, offset_dev=self.offset_dev)
        self.offset += 1
        return out

    def reseed(self, seed, offset=0):
        self.seed, self.offset = int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset)
        if self.offset_dev is not None:
            self.offset_dev.fill_(self.offset)
