"""NCE loss with a fixed partition constant (reference: criterions/nce.py:14-58)."""
import torch
from torch import nn
import torch.distributed as dist

from avid_hip import ops


class NCECriterion(nn.Module):
    """``forward(scores_pos [bs,P], scores_neg [bs,K]) -> 0-d loss`` (autograd-capable).

    ``avg_exp_score`` (buffer, init -1) is Z: computed on the first call as mean(exp(scores_neg)),
    averaged over ranks, then frozen (nce.py:21-36).  It stays on the device — the reference's
    ``if self.avg_exp_score > 0`` host sync happens here once, not every step.  Divergence: the
    buffer keeps shape () instead of silently becoming (1,) after the first call.
    """

    def __init__(self, nLem):
        super(NCECriterion, self).__init__()
        self.nLem = nLem
        self.register_buffer('avg_exp_score', torch.tensor(-1.))
        self.distributed = dist.is_available() and dist.is_initialized()
        self._z_ready = None     # unknown until checked once (a loaded checkpoint may carry Z)
        self._register_state_dict_hook(self._reference_shape)

    @staticmethod
    def _reference_shape(module, state_dict, prefix, local_metadata):
        """Checkpoint interchange: once Z is computed the reference's buffer has shape (1,) (nce.py:27-35 assigns the
        gathered mean), before that (); what this module saves has the same shape at the same time."""
        key = prefix + 'avg_exp_score'
        if key in state_dict and module.z_ready():
            state_dict[key] = state_dict[key].reshape(1)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        key = prefix + 'avg_exp_score'
        if key in state_dict and state_dict[key].dim() == 1:      # reference checkpoints store shape (1,)
            state_dict[key] = state_dict[key].reshape(())
        self._z_ready = None
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def z_ready(self):
        """Z is frozen (nce.py:22-24 ``if self.avg_exp_score > 0``).  The host looks at the buffer once — after
        construction or ``load_state_dict`` — and never again."""
        if self._z_ready is None:
            self._z_ready = bool(self.avg_exp_score.item() > 0)
        return self._z_ready

    def compute_partition_function(self, scores_neg):
        """NOTE: takes the raw negative scores (the exp is fused into the reduction kernel)."""
        if self.z_ready():
            return self.avg_exp_score
        with torch.no_grad():
            Z = ops.mean_exp(scores_neg)
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                dist.all_reduce(Z)                      # == mean of the all-gathered batch means (nce.py:29-30)
                Z /= dist.get_world_size()
            self.avg_exp_score.copy_(Z)
        self._z_ready = True
        return self.avg_exp_score

    def forward(self, scores_pos, scores_neg):
        Z = self.compute_partition_function(scores_neg)
        return ops.nce_loss(scores_pos, scores_neg, Z)
