"""AVID memory-bank criterion on gfx950 kernels (reference: criterions/avid.py:20-236)."""
import torch
from torch import nn
import torch.distributed as dist

from avid_hip import ops
from utils.distributed_utils import _gather_from_all
from utils.alias_method import AliasMethod
from criterions.nce import NCECriterion

__all__ = ['AVID']


def _device_of(device):
    if isinstance(device, torch.device):
        return device
    return torch.device('cuda', int(device) if device is not None else torch.cuda.current_device())


def gather_update_records(video_emb, audio_emb, y):
    """The three all_gathers of avid.py:108-111 fused into ONE: a packed [bs, 2D+2] fp32 record
    (the int64 id travels bit-cast as two floats), 1032 B/sample at D = 128."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return video_emb, audio_emb, y
    bs, D = video_emb.shape
    rec = torch.empty((bs, 2 * D + 2), dtype=torch.float32, device=video_emb.device)
    rec[:, :D] = video_emb
    rec[:, D:2 * D] = audio_emb
    rec[:, 2 * D:] = y.contiguous().view(torch.float32).view(bs, 2)
    allrec = _gather_from_all(rec)
    y_all = allrec[:, 2 * D:].contiguous().view(torch.int64).view(-1)
    return allrec[:, :D].contiguous(), allrec[:, D:2 * D].contiguous(), y_all


class AVIDSimilarityMemoryBank(nn.Module):
    def __init__(self, memory_size, embedding_dim, xModal=True, wModal=False, num_negatives=1024, momentum=0.5,
                 device=0):
        super(AVIDSimilarityMemoryBank, self).__init__()
        self.num_negatives = num_negatives
        self.temperature = 0.07
        if not isinstance(momentum, (list, tuple)):
            momentum = [momentum] * 2
        self.momentum = momentum
        self.device = device

        self.multinomial = AliasMethod(torch.ones(memory_size - 1))
        self.xModal = xModal
        self.wModal = wModal

        self.distributed = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if self.distributed else 0
        self.multinomial.seed = (self.multinomial.seed + 0x9E3779B97F4A7C15 * self.rank) & 0xFFFFFFFFFFFFFFFF  # per-rank negatives

        self.init_memory(memory_size, embedding_dim)

    # ------------------------------------------------------------------ forward (avid.py:47-80)
    def forward(self, video_emb, audio_emb, y):
        K = int(self.num_negatives)
        inv_T = 1.0 / self.temperature
        ops.poll_device_errors(y.device)      # an out-of-range id of an earlier step raises IndexError here (no sync)
        video_emb = ops.l2_normalize(video_emb)
        audio_emb = ops.l2_normalize(audio_emb)

        with torch.no_grad():
            idx = self.sample_negatives(y, K)
            rows = torch.cat([y.view(-1, 1), idx], 1)        # column 0 = the positive (self) row

        scores = {}
        if self.xModal:
            s = ops.bank_scores(video_emb, self.view2_mem, rows, inv_T)
            scores['v2a'] = ops.split_scores(s, 1)
            s = ops.bank_scores(audio_emb, self.view1_mem, rows, inv_T)
            scores['a2v'] = ops.split_scores(s, 1)
        if self.wModal:
            s = ops.bank_scores(video_emb, self.view1_mem, rows, inv_T)
            scores['v2v'] = ops.split_scores(s, 1)
            s = ops.bank_scores(audio_emb, self.view2_mem, rows, inv_T)
            scores['a2a'] = ops.split_scores(s, 1)

        # Update memory bank (scores above used the pre-update rows; their snapshot is kept for backward)
        self.update_memory(video_emb.detach(), audio_emb.detach(), y)
        ops.poll_device_errors(y.device)      # starts the (asynchronous) read-back of this step's error word
        return scores

    def fused_ok(self, video_emb, audio_emb):
        """The one-kernel steady-state path (ops.xmodal_fused): cross-modal scores only, 128-d embeddings on the GPU."""
        return (ops.FUSED_CRITERION and self.xModal and not self.wModal and video_emb.is_cuda and video_emb.dim() == 2
                and video_emb.shape[1] == 128 and self.view1_mem.shape[1] == 128
                and video_emb.dtype == torch.float32 and audio_emb.dtype == torch.float32)

    def forward_fused(self, video_emb, audio_emb, y, Z, coeff):
        """avid.py:47-80 + nce.py:38-58 for both cross-modal score sets with the partition constant ``Z`` frozen: one
        kernel for normalize -> gather (both banks, shared indices) -> scores / T -> NCE terms AND their gradient with
        respect to the embeddings, then ONE bank-update launch.  Returns (total loss, losses[4])."""
        K = int(self.num_negatives)
        ops.poll_device_errors(y.device)
        with torch.no_grad():
            idx = self.sample_negatives(y, K)
        ws = getattr(self, "_fused_ws", None)
        key = (y.shape[0], K, video_emb.device)
        if ws is None or self._fused_key != key:
            ws = self._fused_ws = ops.xmodal_fused_workspace(video_emb.device, y.shape[0], K)
            self._fused_key = key
        total, losses, hats = ops.xmodal_fused(video_emb, audio_emb, y, idx, self.view1_mem, self.view2_mem, Z,
                                               1.0 / self.temperature, coeff, ws)
        self.update_memory(hats[0], hats[1], y)
        ops.poll_device_errors(y.device)
        return total, losses

    def sample_negatives(self, y, K):
        """avid.py:82-86 — uniform over [0,N) \\ {y}; draw + "avoid self" in one kernel."""
        bs = y.shape[0]
        if self.multinomial.prob.device != y.device:
            self.multinomial.to(y.device)
        return self.multinomial.draw(bs * K, y=y.contiguous(), per_row=K).view(bs, K)

    def init_memory(self, num_items, embedding_dim):
        dev = _device_of(self.device)
        self.register_buffer('view1_mem', torch.nn.functional.normalize(torch.randn(num_items, embedding_dim), p=2, dim=1))
        self.register_buffer('view2_mem', torch.nn.functional.normalize(torch.randn(num_items, embedding_dim), p=2, dim=1))
        self.view1_mem = self.view1_mem.to(dev)
        self.view2_mem = self.view2_mem.to(dev)
        self.multinomial.to(dev)
        if self.distributed:
            dist.broadcast(self.view1_mem, 0)
            dist.broadcast(self.view2_mem, 0)
            dist.barrier()

    def update_memory(self, video_emb, audio_emb, y):
        """avid.py:103-129: EMA + renormalise the rows of every sample of the GLOBAL batch, both banks.
        Duplicate ids: last occurrence (highest global position) wins — the reference is unordered."""
        video_mom = float(self.momentum[0])
        audio_mom = float(self.momentum[1])
        v_all, a_all, y_all = gather_update_records(video_emb, audio_emb, y)
        with torch.no_grad():
            if self.view1_mem.is_cuda and self.view1_mem.shape[1] <= 512:
                ops.bank_update_pair(self.view1_mem, self.view2_mem, y_all, v_all, a_all, video_mom, audio_mom)
            else:
                ops.bank_update(self.view1_mem, y_all, v_all, video_mom)
                ops.bank_update(self.view2_mem, y_all, a_all, audio_mom)

    def __repr__(self):
        return describe_bank(self)


def describe_bank(bank):
    """What ``str(criterion)`` shows in the training log (utils/main_utils.py:235): sizes and hyper-parameters."""
    rows = [("name", bank._get_name()), ("num_negatives", int(bank.num_negatives)),
            ("momentum", [float(m) for m in bank.momentum[:2]]),
            ("view1_buffer_size", tuple(bank.view1_mem.shape)), ("view2_buffer_size", tuple(bank.view2_mem.shape))]
    return "{ " + ",\n  ".join(f"{k!r}: {v!r}" for k, v in rows) + "}"


def restore_banks_and_partition(criterion, checkpoint):
    """A criterion picks up the memory banks and the partition constant of an earlier run (criterions/avid.py:187-200,
    criterions/avid_cma.py:308-319): both banks verbatim, and ONE Z = the mean of every ``avg_exp_score`` the file holds,
    written into every NCE term of this criterion.  Buffers are overwritten in place (shapes must agree)."""
    saved = torch.load(checkpoint, map_location='cpu')['train_criterion']
    zs = [v.reshape(()) for k, v in saved.items() if 'avg_exp_score' in k]
    with torch.no_grad():
        for name in ('view1_mem', 'view2_mem'):
            dst, src = getattr(criterion.nce_average, name), saved['nce_average.' + name]
            if tuple(dst.shape) != tuple(src.shape):
                raise RuntimeError(f"checkpoint bank {name} has shape {tuple(src.shape)}, this criterion {tuple(dst.shape)}")
            dst.copy_(src)
        if zs:
            z = torch.stack(zs).mean()
            for m in criterion.modules():
                if isinstance(m, NCECriterion):
                    m.avg_exp_score.copy_(z)
                    m._z_ready = None


def combine_losses(terms, groups, tb_log):
    """``terms``: {score-set name: NCE loss}; ``groups``: [(names of the set pair, coefficient)].  Every group's loss is
    the mean of its (up to two) directions, the total their coefficient-weighted sum (criterions/avid.py:216-232,
    criterions/avid_cma.py:338-358), accumulated in the reference's order."""
    total, per_group = None, []
    for names, coeff in groups:
        acc = 0.
        for k in terms:
            if k in names:
                acc = acc + terms[k] / 2.
        per_group.append(acc)
        part = acc * coeff
        total = part if total is None else total + part
    for k, v in terms.items():
        tb_log[f'Loss/{k}'] = v
    return total, per_group


class AVID(nn.Module):
    """AVID criterion — same constructor, ``forward(emb1, emb2, target) -> (loss, tb_log)``,
    ``set_epoch`` and state_dict keys as the reference (avid.py:145-236)."""

    def __init__(self, num_data, embedding_dim, num_negatives=4096, momentum=0.9, xModal_coeff=1., wModal_coeff=0.,
                 checkpoint=None, device=0):
        super(AVID, self).__init__()
        self.nce_average = AVIDSimilarityMemoryBank(
            memory_size=num_data,
            embedding_dim=embedding_dim,
            num_negatives=num_negatives,
            momentum=momentum,
            xModal=xModal_coeff > 0.,
            wModal=wModal_coeff > 0.,
            device=device
        )
        sum_coeff = (xModal_coeff + wModal_coeff)
        self.xModal_coeff = xModal_coeff / sum_coeff
        self.wModal_coeff = wModal_coeff / sum_coeff
        self.criterion = NCECriterion(num_data).to(_device_of(device))

        if checkpoint is not None:
            restore_banks_and_partition(self, checkpoint)

    def forward(self, emb1, emb2, target):
        tb_log = {}
        # steady state (Z frozen after the first batch, nce.py:22-24) with cross-modal scores only: one fused kernel
        if self.nce_average.fused_ok(emb1, emb2) and self.criterion.z_ready():
            total_loss, losses = self.nce_average.forward_fused(emb1, emb2, target, self.criterion.avg_exp_score,
                                                                self.xModal_coeff)
            tb_log['Loss/v2a'], tb_log['Loss/a2v'] = losses[0], losses[1]
            tb_log['Loss/xModal'] = losses[2]
            tb_log['Loss/wModal'] = 0
            return total_loss, tb_log
        scores = self.nce_average(emb1, emb2, target)
        terms = {k: self.criterion(*pair) for k, pair in scores.items()}   # one shared NCECriterion: Z comes from 'v2a'
        total_loss, (xm, wm) = combine_losses(terms, [(('v2a', 'a2v'), self.xModal_coeff), (('v2v', 'a2a'), self.wModal_coeff)],
                                              tb_log)
        tb_log['Loss/xModal'], tb_log['Loss/wModal'] = xm, (wm if torch.is_tensor(wm) else 0)
        return total_loss, tb_log

    def set_epoch(self, epoch):
        pass
