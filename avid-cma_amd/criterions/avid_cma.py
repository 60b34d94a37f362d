"""AVID+CMA criterion on gfx950 kernels (reference: criterions/avid_cma.py:24-364)."""
import torch
from torch import nn
import torch.distributed as dist

from avid_hip import ops
from utils.alias_method import AliasMethod
from criterions.nce import NCECriterion
from criterions.avid import (AVIDSimilarityMemoryBank, _device_of, combine_losses, describe_bank,
                             restore_banks_and_partition)

__all__ = ['AVID_CMA']

_KINDS = {'consensus': 0, 'union': 1, 'video': 2, 'audio': 3}


class CMASampler:
    """Cross-modal-agreement search (avid_cma.py:24-123).

    The reference fans 16-query jobs out to one child process per GPU through mp.Queues while the
    other ranks wait in a barrier.  Here every rank searches its own contiguous shard of query rows
    in-process (each rank already holds both banks) and the shards are all-gathered.
    """

    def __init__(self, video_mem, audio_mem, sampling_args):
        self.video_mem = video_mem
        self.audio_mem = audio_mem
        self.sampling_args = sampling_args
        if sampling_args['type'] not in _KINDS:
            raise ValueError
        # diagnostics: query batches whose candidate lists overflowed the threshold filter (exact scan redid them)
        self.fallback_batches = torch.zeros((), dtype=torch.int32, device=video_mem.device) if video_mem.is_cuda else None

    def sample_range(self, q0, q1, batch=1024):
        from avid_hip import topk
        return topk.cma_topk(self.video_mem, self.audio_mem, q0, q1, self.sampling_args['pos_k'],
                             _KINDS[self.sampling_args['type']], batch, fallbacks=self.fallback_batches)

    def sample(self):
        N = self.video_mem.shape[0]
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
        per = (N + world - 1) // world
        q0, q1 = min(rank * per, N), min((rank + 1) * per, N)
        local = self.sample_range(q0, q1)
        if world == 1:
            return local
        padded = torch.zeros((per, local.shape[1]), dtype=local.dtype, device=local.device)
        padded[:q1 - q0] = local
        out = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(out, padded)
        return torch.cat(out, 0)[:N]


class AVIDSimilarityPositiveExpansion(AVIDSimilarityMemoryBank):
    def __init__(self, memory_size, embedding_dim, xModalInst=True, wModalInst=False, xModalPos=False,
                 wModalPos=True, num_negatives=1024, num_negatives_within=None, sampling_args=None, momentum=0.5,
                 device=0):
        super().__init__(memory_size=memory_size, embedding_dim=embedding_dim, xModal=xModalInst, wModal=wModalInst,
                         num_negatives=num_negatives, momentum=momentum, device=device)
        self.num_negatives_within = num_negatives_within
        self.multinomial = AliasMethod(torch.ones(memory_size - sampling_args['pos_k']))
        self.multinomial.seed = (self.multinomial.seed + 0x9E3779B97F4A7C15 * self.rank) & 0xFFFFFFFFFFFFFFFF
        self.multinomial.to(_device_of(device))
        self.sampling_args = sampling_args

        self.xModalInst = xModalInst
        self.wModalInst = wModalInst
        self.xModalPos = xModalPos
        self.wModalPos = wModalPos

    def forward(self, video_emb, audio_emb, y):
        """avid_cma.py:150-194."""
        inv_T = 1.0 / self.temperature
        bs = y.shape[0]
        ops.poll_device_errors(y.device)
        video_emb = ops.l2_normalize(video_emb)
        audio_emb = ops.l2_normalize(audio_emb)

        with torch.no_grad():
            pos_idx, neg_idx = self.memory_sampling(y)
            P = pos_idx.shape[1]
            rows = torch.cat([y.view(-1, 1), pos_idx, neg_idx], 1)      # [self | P positives | K negatives]

        s_v2a = s_a2v = None
        if self.xModalInst or self.wModalInst or self.xModalPos:
            s_v2a = ops.bank_scores(video_emb, self.view2_mem, rows, inv_T)   # video emb vs audio bank
            s_a2v = ops.bank_scores(audio_emb, self.view1_mem, rows, inv_T)   # audio emb vs video bank

        scores = {}
        neg = slice(1 + P, None)
        if self.xModalInst:
            scores['inst-v2a'] = [s_v2a[:, :1], s_v2a[:, neg]]
            scores['inst-a2v'] = [s_a2v[:, :1], s_a2v[:, neg]]
        if self.wModalInst:   # reference quirk kept: overwrites the same keys with the same cross-modal scores (:175-177)
            scores['inst-v2a'] = [s_v2a[:, :1], s_v2a[:, neg]]
            scores['inst-a2v'] = [s_a2v[:, :1], s_a2v[:, neg]]
        if self.xModalPos:
            scores['pos-v2a'] = [s_v2a[:, 1:1 + P], s_v2a[:, neg]]
            scores['pos-a2v'] = [s_a2v[:, 1:1 + P], s_a2v[:, neg]]
        if self.wModalPos:
            Kw = neg_idx.shape[1] if self.num_negatives_within is None else int(self.num_negatives_within)
            wrows = rows[:, 1:1 + P + Kw].contiguous()                  # [P positives | first Kw negatives]
            s_v2v = ops.bank_scores(video_emb, self.view1_mem, wrows, inv_T)
            s_a2a = ops.bank_scores(audio_emb, self.view2_mem, wrows, inv_T)
            scores['pos-v2v'] = ops.split_scores(s_v2v, P)
            scores['pos-a2a'] = ops.split_scores(s_a2a, P)

        self.update_memory(video_emb.detach(), audio_emb.detach(), y)
        ops.poll_device_errors(y.device)
        return scores

    def fused_ok(self, video_emb, audio_emb):
        """The one-kernel steady-state path (ops.cma_fused): the stock term set — cross-modal instance + within-modal
        positives (InstX-...-PosW-..., the config of configs/main/avid-cma) — on 128-d fp32 embeddings on the GPU."""
        return (ops.FUSED_CRITERION and self.xModalInst and self.wModalPos and not self.wModalInst and not self.xModalPos
                and self.sampling_args['pos_k'] > 0 and video_emb.is_cuda and video_emb.dim() == 2
                and video_emb.shape[1] == 128 and self.view1_mem.shape[1] == 128
                and video_emb.dtype == torch.float32 and audio_emb.dtype == torch.float32)

    def forward_fused(self, video_emb, audio_emb, y, Z, coeff_inst, coeff_pos):
        """forward() + the four NCE terms (nce.py:38-58, Z frozen) + their gradient in one kernel, then ONE bank-update
        launch.  Returns (total loss, losses[8])."""
        ops.poll_device_errors(y.device)
        with torch.no_grad():
            pos_idx, neg_idx = self.memory_sampling(y)
        P, K = pos_idx.shape[1], neg_idx.shape[1]
        Kw = K if self.num_negatives_within is None else min(int(self.num_negatives_within), K)
        ws = getattr(self, "_fused_ws", None)
        key = (y.shape[0], P, K, video_emb.device)
        if ws is None or self._fused_key != key:
            ws = self._fused_ws = ops.cma_fused_workspace(video_emb.device, y.shape[0], P, K)
            self._fused_key = key
        total, losses, hats = ops.cma_fused(video_emb, audio_emb, y, pos_idx, neg_idx, self.view1_mem, self.view2_mem, Z,
                                            1.0 / self.temperature, Kw, coeff_inst, coeff_pos, ws)
        self.update_memory(hats[0], hats[1], y)
        ops.poll_device_errors(y.device)
        return total, losses

    def memory_sampling(self, y):
        """avid_cma.py:196-209: positives = positive_set[y]; negatives skip the (sorted) positives."""
        bs = y.shape[0]
        if self.multinomial.prob.device != y.device:
            self.multinomial.to(y.device)
        rand_idx = self.multinomial.draw(bs * self.num_negatives).view(bs, -1)
        return ops.cma_negatives(self.positive_set, y, rand_idx)

    def find_correspondences(self):
        """avid_cma.py:211-229 — every rank searches its shard; result all-gathered (no rank-0 bottleneck)."""
        if self.sampling_args['pos_k'] <= 0:
            return
        positive_set = CMASampler(self.view1_mem, self.view2_mem, self.sampling_args).sample()
        positive_set = positive_set.int().to(self.view1_mem.device)
        old = getattr(self, 'positive_set', None)
        if old is not None and old.shape == positive_set.shape and old.device == positive_set.device:
            old.copy_(positive_set)      # in place: a captured hipGraph of the step keeps reading this buffer
        else:
            self.register_buffer('positive_set', positive_set)
        if self.distributed:
            dist.barrier()

    def __repr__(self):
        return describe_bank(self)


class AVID_CMA(nn.Module):
    """Same constructor / forward / set_epoch / state_dict keys as the reference (avid_cma.py:245-364)."""

    def __init__(self, num_data, embedding_dim, num_negatives=1024, num_negatives_within=None, momentum=0.5,
                 xModalInstCoeff=1., wModalInstCoeff=0., xModalPosCoeff=0., wModalPosCoeff=1., sampling_args=None,
                 checkpoint=None, resample_freq=-1, device=0):
        super(AVID_CMA, self).__init__()
        self.nce_average = AVIDSimilarityPositiveExpansion(
            memory_size=num_data,
            embedding_dim=embedding_dim,
            num_negatives=num_negatives,
            num_negatives_within=num_negatives_within,
            momentum=momentum,
            xModalInst=xModalInstCoeff > 0.,
            xModalPos=xModalPosCoeff > 0.,
            wModalInst=wModalInstCoeff > 0.,
            wModalPos=wModalPosCoeff > 0.,
            sampling_args=sampling_args,
            device=device
        )
        sum_coeff = xModalInstCoeff + wModalInstCoeff + xModalPosCoeff + wModalPosCoeff
        self.xModalInstCoeff = xModalInstCoeff / sum_coeff
        self.wModalInstCoeff = wModalInstCoeff / sum_coeff
        self.xModalPosCoeff = xModalPosCoeff / sum_coeff
        self.wModalPosCoeff = wModalPosCoeff / sum_coeff

        self.criterion = NCECriterion(num_data).to(_device_of(device))

        if checkpoint is not None:       # an AVID run's banks and Z (avid_cma.py:308-319)
            restore_banks_and_partition(self, checkpoint)

        self.resample_freq = resample_freq
        self.nce_average.find_correspondences()

    def forward(self, emb1, emb2, target):
        tb_log = {}
        # steady state (Z frozen after the first batch, nce.py:22-24) with the stock term set: one fused kernel
        if self.nce_average.fused_ok(emb1, emb2) and self.criterion.z_ready():
            total_loss, losses = self.nce_average.forward_fused(emb1, emb2, target, self.criterion.avg_exp_score,
                                                                self.xModalInstCoeff, self.wModalPosCoeff)
            for k, name in enumerate(('inst-v2a', 'inst-a2v', 'pos-v2v', 'pos-a2a')):
                tb_log[f'Loss/{name}'] = losses[k]
            return total_loss, tb_log
        scores = self.nce_average(emb1, emb2, target)
        terms = {k: self.criterion(*pair) for k, pair in scores.items()}
        total_loss, _ = combine_losses(terms, [(('inst-v2a', 'inst-a2v'), self.xModalInstCoeff),
                                               (('inst-v2v', 'inst-a2a'), self.wModalInstCoeff),
                                               (('pos-v2a', 'pos-a2v'), self.xModalPosCoeff),
                                               (('pos-v2v', 'pos-a2a'), self.wModalPosCoeff)], tb_log)
        return total_loss, tb_log

    def set_epoch(self, epoch):
        if self.resample_freq > 0 and epoch > 0 and epoch % self.resample_freq == 0:
            self.nce_average.find_correspondences()
