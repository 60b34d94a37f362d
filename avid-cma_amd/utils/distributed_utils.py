"""reference: utils/distributed_utils.py:12-19."""
import torch
from torch import distributed as dist


def _gather_from_all(tensor):
    """all_gather + cat(dim 0) in rank order (RCCL over xGMI on the GPU box, gloo in CPU tests)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tensor
    tensor = tensor.contiguous()
    out = torch.empty((dist.get_world_size() * tensor.shape[0],) + tuple(tensor.shape[1:]), dtype=tensor.dtype,
                      device=tensor.device)
    dist.all_gather_into_tensor(out, tensor) if hasattr(dist, "all_gather_into_tensor") and tensor.is_cuda else \
        dist.all_gather(list(out.chunk(dist.get_world_size(), 0)), tensor)
    return out
