"""R(2+1)D video encoder on gfx950 kernels (reference: models/video.py:12-54)."""
import os
import torch.nn as nn

from avid_hip import ops
from .network_blocks import BasicR2P1DBlock, BatchNormCL, ConvCL, MaxPoolHW3S2, no_bn_handover

_FUSE_STEM_TAIL = os.environ.get("AVID_FUSE_STEM_TAIL", "1") == "1"


def _stem_conv(conv, x):
    """Stem convolution + (when the layer can) the BatchNorm partial sums of its output from the epilogue."""
    from .network_blocks import _FUSE_BN_STATS
    if not _FUSE_BN_STATS:
        return conv(x), None
    return conv(x, bn_stats=True)

__all__ = ["R2Plus1D"]

_STAGES = {10: (1, 1, 1, 1), 18: (2, 2, 2, 2), 34: (3, 4, 6, 3)}


def _ncdhw(x):
    """channels-last [B,T,H,W,C] -> logical [B,C,T,H,W] view (no copy)."""
    return x.permute(0, 4, 1, 2, 3)


class R2Plus1D(nn.Module):
    """``forward(x [B,3,T,H,W], return_embs=False) -> [B,512,1,1,1]`` exactly as the reference.

    Full (3,7,7) Conv3d stem (read straight from the NCDHW input), no mid-channel adjustment,
    depth 10 / 18 / 34 (models/video.py:26-40).
    """

    def __init__(self, depth=18):
        super().__init__()
        self.conv1 = nn.Sequential(
            ConvCL(3, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3), channel_first=True),
            BatchNormCL(64),
            nn.ReLU(inplace=True),
            MaxPoolHW3S2(),
        )
        n = _STAGES[depth]
        chans = [(64, 64), (64, 128), (128, 256), (256, 512)]
        for name, (cin, cout), nb, first in zip(["conv2x", "conv3x", "conv4x", "conv5x"], chans, n,
                                                [False, True, True, True]):
            blocks = []
            for b in range(nb):
                stride = (2, 2, 2) if (first and b == 0) else (1, 1, 1)
                blocks.append(BasicR2P1DBlock(cin if b == 0 else cout, cout, stride=stride))
            setattr(self, name, blocks[0] if depth == 10 else nn.Sequential(*blocks))
        self.pool = nn.AdaptiveMaxPool3d((1, 1, 1))   # module-tree parity; computed by ops.global_maxpool
        self.out_dim = 512

    def forward(self, x, return_embs=False):
        if return_embs:        # intermediate activations leave the module: no single-consumer hand-over
            with no_bn_handover():
                return self._forward(x, True)
        return self._forward(x, False)

    def _forward(self, x, return_embs):
        conv, bn = self.conv1[0], self.conv1[1]
        if self.training and x.is_cuda and _FUSE_STEM_TAIL:
            # BN + ReLU + max-pool in one pass over the 411 MB stem activation (bs 64)
            # (its batch statistics come out of the stem convolution's epilogue when the layer can: _FUSE_BN_STATS)
            y, part = _stem_conv(conv, x.contiguous())
            x_c1 = ops.bn_relu_maxpool(y, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                       bn.momentum, bn.eps, bn.num_batches_tracked, partials=part)
        else:
            x_c1 = self.conv1[3](bn(conv(x.contiguous()), relu=True))
        hook = getattr(self, "_stage_hook", None)     # (stage index, issued so far) -> AV_Wrapper starts the audio tower
        if hook is not None:
            hook(1)
        x_b1 = self.conv2x(x_c1)
        if hook is not None:
            hook(2)
        x_b2 = self.conv3x(x_b1)
        if hook is not None:
            hook(3)
        x_b3 = self.conv4x(x_b2)
        if hook is not None:
            hook(4)
        x_b4 = self.conv5x(x_b3)
        pooled = ops.global_maxpool(x_b4)
        x_pool = pooled.view(pooled.shape[0], pooled.shape[1], 1, 1, 1)
        if return_embs:
            return {"conv1": _ncdhw(x_c1), "conv2x": _ncdhw(x_b1), "conv3x": _ncdhw(x_b2), "conv4x": _ncdhw(x_b3),
                    "conv5x": _ncdhw(x_b4), "pool": x_pool}
        return x_pool
