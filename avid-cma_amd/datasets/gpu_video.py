"""GPU-side tail of the reference's video preprocessing (datasets/preprocessing.py:45-48):
``volume_transforms.ClipToTensor()`` + ``tensor_transforms.Normalize(mean, std)`` for a whole batch of decoded,
already cropped / augmented uint8 clips in ONE HBM-bound kernel (avid_clip_normalize), bit-identical to the CPU
transform.  With it the DataLoader workers hand over uint8 frames (a quarter of the bytes of the fp32 tensor over
PCIe) and the float conversion happens on the training stream right before ``model(video, audio)``."""
import torch

from avid_hip import ops

__all__ = ["ClipToTensorNormalize"]


class ClipToTensorNormalize:
    """``__call__(frames)``: ``frames`` uint8 ``[B, T, H, W, 3]`` (or one clip ``[T, H, W, 3]``) on the GPU ->
    fp32 ``[B, 3, T, H, W]`` (``[3, T, H, W]``): ((u / 255) - mean) / std per channel."""

    def __init__(self, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
        self.mean, self.std = tuple(mean), tuple(std)

    def __call__(self, frames):
        single = frames.dim() == 4
        if single:
            frames = frames.unsqueeze(0)
        out = ops.clip_normalize(frames.contiguous(), self.mean, self.std)
        return out[0] if single else out

    def __repr__(self):
        return f"ClipToTensorNormalize(mean={self.mean}, std={self.std})"
