"""Building blocks of the two encoders on the gfx950 kernels (reference: models/network_blocks.py).

Modules keep the reference's attribute names, parameter shapes and state_dict keys; internally
activations are channels-last ``[B,T,H,W,C]`` and every op is a hand-written HIP kernel
(avid_hip.ops).  Conv+residual-add and BN+ReLU are fused.
"""
import math
import os

import torch
import torch.nn as nn

from avid_hip import ops


class ConvCL(nn.Module):
    """nn.Conv3d / nn.Conv2d (bias=False) replacement.

    ``weight`` has the reference's logical shape ``[Cout, Cin, *k]`` and torch's default init
    (kaiming_uniform(a=sqrt(5)), as nn.Conv3d.reset_parameters) but channels-last memory.
    """

    def __init__(self, in_planes, out_planes, kernel_size, stride, padding, channel_first=False):
        super().__init__()
        self.in_channels, self.out_channels = in_planes, out_planes
        self.kernel_size, self.ndim = tuple(kernel_size), len(kernel_size)
        pad3 = lambda v: (1,) * 0 + ((v,) * 3 if isinstance(v, int) else tuple(v))  # noqa: E731
        k3 = (1,) * (3 - self.ndim) + self.kernel_size
        s = pad3(stride) if not isinstance(stride, int) else (stride,) * self.ndim
        p = pad3(padding) if not isinstance(padding, int) else (padding,) * self.ndim
        self.stride3 = (1,) * (3 - len(s)) + tuple(s)
        self.padding3 = (0,) * (3 - len(p)) + tuple(p)
        self.k3 = k3
        self.channel_first = channel_first
        self.weight = nn.Parameter(ops.make_weight(out_planes, in_planes, *self.kernel_size))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def forward(self, x, addend=None, bn_stats=False, tap=False, bn_src=None, res=None):
        """``bn_stats=True``: also return the BatchNorm partial sums of the output (for the BN that follows);
        ``tap=True``: also return an alias of ``x`` for a second consumer; ``bn_src`` / ``res``: see ``ops.conv_cl``."""
        return ops.conv_cl(x, self.weight, self.stride3, self.padding3, addend=addend,
                           channel_first=self.channel_first, bn_stats=bn_stats, tap=tap, bn_src=bn_src, res=res)

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, "
                f"stride={self.stride3[3 - self.ndim:]}, padding={self.padding3[3 - self.ndim:]}, bias=False")


class BatchNormCL(nn.Module):
    """nn.BatchNorm3d/2d replacement over channels-last activations, optional fused ReLU.

    Same parameters / buffers / defaults (eps 1e-5, momentum 0.1, affine, track_running_stats).
    """

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, x, relu=False, partials=None, src=None):
        return ops.batch_norm_cl(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                                 self.momentum, self.eps, relu,
                                 self.num_batches_tracked if self.training else None,   # bumped inside the kernel
                                 partials, src)

    def extra_repr(self):
        return f"{self.num_features}, eps={self.eps}, momentum={self.momentum}"


_FUSE_BN_STATS = os.environ.get("AVID_FUSE_BN_STATS", "1") == "1"


_HANDOVER = [True]


class no_bn_handover:
    """Context: activations produced inside are NOT marked as single-consumer BatchNorm outputs (``return_embs``
    hands intermediate activations to the caller, who may put them in the loss: then the next convolution's
    input gradient is only part of the BatchNorm output's gradient and the hand-over would be wrong)."""

    def __enter__(self):
        self.prev, _HANDOVER[0] = _HANDOVER[0], False

    def __exit__(self, *exc):
        _HANDOVER[0] = self.prev


_FUSE_RES = os.environ.get("AVID_FUSE_RES", "1") == "1"


def _conv_bn(conv, bn, x, addend=None, tap=False, sole=True, res=None):
    """ReLU(bn(conv(x) [+ addend])).  In training the conv epilogue hands the BatchNorm its batch statistics
    as partial sums, so the BN does not re-read the activation for them.  ``tap``: also return an alias of x
    whose gradient is folded into this conv's input-gradient kernel (the residual branch).
    ``sole``: this conv (with its tap) is the only consumer of x; if x is itself the output of a ``_conv_bn``, the
    backward partial sums of THAT BatchNorm are then made by this conv's input-gradient kernel (ops.BnSource)."""
    train = bn.training and x.is_cuda
    fuse = train and _FUSE_BN_STATS
    src_in = getattr(x, "_avid_bn_src", None) if (sole and train and torch.is_grad_enabled()) else None
    out = conv(x, addend=addend, bn_stats=fuse, tap=tap, bn_src=src_in, res=res)
    src_out = ops.BnSource(None, None, True) if (train and torch.is_grad_enabled() and _HANDOVER[0]) else None
    if not (fuse or tap or res is not None):
        h = bn(out, relu=True, src=src_out)
    else:
        h = bn(out[0], relu=True, partials=out[1] if fuse else None, src=src_out)
    if src_out is not None:
        h._avid_bn_src = src_out          # read by the next _conv_bn that takes h as its input
    return (h, out[-1]) if (tap or res is not None) else h


class Basic2DBlock(nn.Module):
    """reference models/network_blocks.py:13-27 — ReLU(bn1(conv1)) -> ReLU(bn2(conv2)); no residual."""

    def __init__(self, in_planes, out_planes, stride=(1, 1)):
        super().__init__()
        if isinstance(stride, int):
            stride = (stride, stride)
        self.conv1 = ConvCL(in_planes, out_planes, (3, 3), stride, (1, 1))
        self.bn1 = BatchNormCL(out_planes)
        self.conv2 = ConvCL(out_planes, out_planes, (3, 3), (1, 1), (1, 1))
        self.bn2 = BatchNormCL(out_planes)
        self.relu = nn.ReLU(inplace=True)      # kept for module-tree parity; fused into the BN kernels

    def forward(self, x):
        x = _conv_bn(self.conv1, self.bn1, x)
        x = _conv_bn(self.conv2, self.bn2, x)
        return x


class BasicR2P1DBlock(nn.Module):
    """reference models/network_blocks.py:30-60.

    spt(1,3,3) -> BN -> ReLU -> tmp(3,1,1) -> BN -> ReLU -> spt -> BN -> ReLU -> tmp ; the residual
    (1x1x1 strided ``res_conv`` or identity) is added inside tmp_conv2's epilogue; out = ReLU(out_bn(.)).
    """

    def __init__(self, in_planes, out_planes, stride=(1, 1, 1)):
        super().__init__()
        spt_stride = (1, stride[1], stride[2])
        tmp_stride = (stride[0], 1, 1)
        self.spt_conv1 = ConvCL(in_planes, out_planes, (1, 3, 3), spt_stride, (0, 1, 1))
        self.spt_bn1 = BatchNormCL(out_planes)
        self.tmp_conv1 = ConvCL(out_planes, out_planes, (3, 1, 1), tmp_stride, (1, 0, 0))
        self.tmp_bn1 = BatchNormCL(out_planes)
        self.spt_conv2 = ConvCL(out_planes, out_planes, (1, 3, 3), (1, 1, 1), (0, 1, 1))
        self.spt_bn2 = BatchNormCL(out_planes)
        self.tmp_conv2 = ConvCL(out_planes, out_planes, (3, 1, 1), (1, 1, 1), (1, 0, 0))
        self.out_bn = BatchNormCL(out_planes)
        self.relu = nn.ReLU(inplace=True)
        if in_planes != out_planes or any(s != 1 for s in stride):
            self.res = True
            self.res_conv = ConvCL(in_planes, out_planes, (1, 1, 1), stride, (0, 0, 0))
        else:
            self.res = False

    def _res_fusable(self, x):
        """The residual convolution can ride inside spt_conv1's op (its input gradient stays on the sub-sampled grid
        and is added in spt_conv1's strided dgrad): strides of 1 / 2, spt_conv1 itself strided, 32-bit offsets."""
        if not self.res:
            return False
        rs, ss = self.res_conv.stride3, self.spt_conv1.stride3
        return (_FUSE_RES and any(v == 2 for v in rs) and all(v in (1, 2) for v in rs)
                and any(v == 2 for v in ss) and all(v in (1, 2) for v in ss) and x.numel() * 4 < (1 << 31))

    def forward(self, x):
        tap = x.is_cuda and x.requires_grad and torch.is_grad_enabled()
        if tap and self._res_fusable(x):
            h, x_res = _conv_bn(self.spt_conv1, self.spt_bn1, x, res=(self.res_conv.weight, self.res_conv.stride3))
            h = _conv_bn(self.tmp_conv1, self.tmp_bn1, h)
            h = _conv_bn(self.spt_conv2, self.spt_bn2, h)
            return _conv_bn(self.tmp_conv2, self.out_bn, h, addend=x_res)
        if tap:     # the residual branch reads an alias of x: its gradient is added inside spt_conv1's dgrad
            h, x = _conv_bn(self.spt_conv1, self.spt_bn1, x, tap=True)
        else:       # (x also feeds the residual branch directly: spt_conv1 is not its only consumer)
            h = _conv_bn(self.spt_conv1, self.spt_bn1, x, sole=False)
        h = _conv_bn(self.tmp_conv1, self.tmp_bn1, h)
        h = _conv_bn(self.spt_conv2, self.spt_bn2, h)
        x_res = self.res_conv(x) if self.res else x
        return _conv_bn(self.tmp_conv2, self.out_bn, h, addend=x_res)


class MaxPoolHW3S2(nn.Module):
    """nn.MaxPool3d((1,3,3),(1,2,2),(0,1,1)) over channels-last activations (models/video.py:23)."""

    def forward(self, x):
        return ops.maxpool_hw3s2(x)
