"""Conv2D audio encoder on gfx950 kernels (reference: models/audio.py:15-44)."""
import torch.nn as nn

from avid_hip import ops
from .network_blocks import Basic2DBlock, BatchNormCL, ConvCL, _conv_bn, no_bn_handover

__all__ = ["Conv2D"]


def _nchw(x):
    """channels-last [B,1,H,W,C] -> logical [B,C,H,W] view."""
    return x[:, 0].permute(0, 3, 1, 2)


class Conv2D(nn.Module):
    """``forward(x [B,1,T,F], return_embs=False) -> [B,512,1,1]``; depth must be 10 (audio.py:18)."""

    def __init__(self, depth=10):
        super().__init__()
        assert depth == 10
        self.conv1 = nn.Sequential(
            ConvCL(1, 64, (7, 7), (2, 2), (3, 3), channel_first=True),
            BatchNormCL(64),
            nn.ReLU(inplace=True),
        )
        self.block1 = Basic2DBlock(64, 64, stride=(2, 2))
        self.block2 = Basic2DBlock(64, 128, stride=(2, 2))
        self.block3 = Basic2DBlock(128, 256, stride=(2, 2))
        self.block4 = Basic2DBlock(256, 512)
        self.pool = nn.AdaptiveMaxPool2d((1, 1))
        self.out_dim = 512

    def forward(self, x, return_embs=False):
        if return_embs:        # intermediate activations leave the module: no single-consumer hand-over
            with no_bn_handover():
                return self._forward(x, True)
        return self._forward(x, False)

    def _forward(self, x, return_embs):
        x5 = x.contiguous().unsqueeze(2)            # [B,1,1,H,W]: 2-D conv == 3-D conv with T = kt = 1
        x_c1 = _conv_bn(self.conv1[0], self.conv1[1], x5)   # stem conv + BN(+ReLU), statistics from the conv epilogue
        x_b1 = self.block1(x_c1)
        x_b2 = self.block2(x_b1)
        x_b3 = self.block3(x_b2)
        x_b4 = self.block4(x_b3)
        pooled = ops.global_maxpool(x_b4)
        x_pool = pooled.view(pooled.shape[0], pooled.shape[1], 1, 1)
        if return_embs:
            return {"conv2x": _nchw(x_b1), "conv3x": _nchw(x_b2), "conv4x": _nchw(x_b3), "conv5x": _nchw(x_b4),
                    "pool": x_pool}
        return x_pool
