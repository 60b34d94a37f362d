"""Two-tower wrapper + projection heads (reference: models/av_wrapper.py:17-76)."""
import math

import torch
import torch.nn as nn

from avid_hip import ops

__all__ = ["av_wrapper"]


class LinearCL(nn.Module):
    """nn.Linear replacement (same ``weight [out,in]`` / ``bias`` and default init); optional fused ReLU."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_features)
        nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, relu=False):
        return ops.linear(x, self.weight, self.bias, relu)

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}, bias=True"


class Head(nn.Module):
    """Linear(+ReLU) x (n-1), Linear — av_wrapper.py:17-33.  ``projection`` keeps indices 0,2,4."""

    def __init__(self, input_dim, proj_dims):
        super().__init__()
        if not isinstance(proj_dims, list):
            proj_dims = [proj_dims]
        projection = []
        for i, d in enumerate(proj_dims):
            projection += [LinearCL(input_dim, d)]
            input_dim = d
            if i < len(proj_dims) - 1:
                projection += [nn.ReLU(inplace=True)]
        self.projection = nn.Sequential(*projection)
        self.out_dim = proj_dims[-1]

    def forward(self, x):
        mods = list(self.projection)
        i = 0
        while i < len(mods):
            fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            x = mods[i](x, relu=fuse)
            i += 2 if fuse else 1
        return x


class AV_Wrapper(nn.Module):
    def __init__(self, video_model, audio_model, proj_dim=128):
        super().__init__()
        self.video_model = video_model
        self.audio_model = audio_model
        self.use_linear_proj = proj_dim is not None
        if proj_dim is not None:
            self.video_proj = Head(video_model.out_dim, proj_dim)
            self.audio_proj = Head(audio_model.out_dim, proj_dim)
            self.out_dim = self.video_proj.out_dim
        else:
            self.out_dim = video_model.out_dim
        self._init_flat_buffers()

    # ---- DistributedDataParallel copies rank 0's buffers to every rank before EVERY forward (utils/main_utils.py:112,
    # broadcast_buffers=True): for this model 126 BatchNorm tensors flattened, broadcast and copied back one by one —
    # 2 ms of host time per step in the reference's loop, which waits for loss.item() and therefore sees it.  The running
    # statistics live in ONE flat buffer instead (`_bn_flat`, not part of the state_dict); the BatchNorm modules' own
    # buffers are views of it, keep their names, shapes and state_dict keys, and are listed in DDP's ignore set, so DDP
    # broadcasts one tensor.  (`num_batches_tracked` advances identically on every rank and is ignored as well.)
    def _init_flat_buffers(self):
        named = [(n, b) for n, b in self.named_buffers() if n != "_bn_flat"]
        floats = [(n, b) for n, b in named if b.is_floating_point()]
        self._bn_layout, off = [], 0
        for n, b in floats:
            self._bn_layout.append((n, off, b.numel()))
            off += (b.numel() + 3) // 4 * 4
        if off:
            self.register_buffer("_bn_flat", torch.zeros(off), persistent=False)
            self._ddp_params_and_buffers_to_ignore = [n for n, _ in named]
        self._bn_seated = None

    def _apply(self, fn, *args, **kwargs):
        # `.to()` / `.cuda()` / `.float()` re-create every buffer: seat them again at once, so that whoever looks at
        # `_bn_flat` next — DistributedDataParallel's construction-time `_sync_module_states`, which skips the ignored
        # per-layer buffers and broadcasts this one — finds the real statistics in it, not zeros
        out = super()._apply(fn, *args, **kwargs)
        if "_bn_layout" in self.__dict__:
            self._bn_seated = None
            self._seat_flat_buffers()
        return out

    def _seat_flat_buffers(self):
        """Make every floating-point BatchNorm buffer a view of `_bn_flat` (again: `.to()` / `.cuda()` re-create buffers)."""
        flat = getattr(self, "_bn_flat", None)
        if flat is None:
            return
        key = (flat.data_ptr(), flat.device)
        if self._bn_seated == key:
            return
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
        mods = dict(self.named_modules())
        with torch.no_grad():
            for n, off, cnt in self._bn_layout:
                owner, _, leaf = n.rpartition(".")
                b = getattr(mods[owner], leaf)
                if b.device != flat.device:
                    return                                   # (model and flat buffer on different devices: leave as is)
                if not (lo <= b.data_ptr() < hi):
                    view = flat[off:off + cnt].view(b.shape)
                    view.copy_(b)
                    b.data = view
        self._bn_seated = key

    # Run the (small) audio tower on a side stream under the video tower's tail waves.  OFF unless the build's
    # own engine drives the step (avid_hip.parallel.TrainStep switches it on): its gradient buckets know which
    # stream produced each gradient.  torch's DistributedDataParallel reducer does not — it orders a bucket's
    # all-reduce after the stream of the LAST gradient hook only, so a bucket mixing the two towers' parameters
    # could be reduced before the other stream's gradient has landed.
    overlap_towers = False

    def _audio(self, audio):
        audio_emb = self.audio_model(audio)
        audio_emb = audio_emb.view(audio_emb.shape[0], audio_emb.shape[1])
        if self.use_linear_proj:
            audio_emb = self.audio_proj(audio_emb)
        return audio_emb

    def forward(self, video, audio):
        self._seat_flat_buffers()
        # A training step through the stock module tree runs as two compiled launch programs (avid_hip/plan.py: one C
        # call per pass, one autograd node for the whole model); anything else — evaluation, hooked modules, frozen
        # parameters, return_embs on a tower — takes the per-layer path below.
        if video.is_cuda and self.training:
            from avid_hip import plan
            out = plan.run(self, video, audio)
            if out is not None:
                return out
        side = None
        capturing = audio.is_cuda and torch.cuda.is_current_stream_capturing()
        if self.overlap_towers and audio.is_cuda and (ops.OVERLAP_IN_CAPTURE or not capturing):
            # The towers are independent until the criterion.  autograd replays each op's backward on the
            # stream its forward ran on, so the audio backward overlaps the video backward as well.
            main = torch.cuda.current_stream()
            side = ops.side_stream(audio.device, 1)
            out = []

            def start_audio(stage):
                # the audio tower starts once the video tower has passed stage AUDIO_AFTER (0: at once): its ~150
                # small kernels then run beside the video tower's small late layers instead of the full-chip stem
                # and conv2x kernels
                if stage >= AUDIO_AFTER and not out:
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        out.append(self._audio(audio))

            start_audio(0)
            self.video_model._stage_hook = start_audio
        try:
            video_emb = self.video_model(video)
        finally:
            if side is not None:
                self.video_model._stage_hook = None
        if side is not None:
            start_audio(99)
            audio_emb = out[0]
        video_emb = video_emb.view(video_emb.shape[0], video_emb.shape[1])
        if self.use_linear_proj:
            video_emb = self.video_proj(video_emb)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
            if not capturing:
                audio_emb.record_stream(torch.cuda.current_stream())
        else:
            audio_emb = self._audio(audio)
        return video_emb, audio_emb


AUDIO_AFTER = 1   # the audio tower starts behind the video stem: +0.4 % (stage 0: 4833, 1: 4854, 2: 4830, 3: 4850, 4: 4854 clips/s)


def av_wrapper(video_backbone, video_backbone_args, audio_backbone, audio_backbone_args, proj_dim=128,
               checkpoint=None):
    """Factory with the reference signature (av_wrapper.py:64); backbones resolved by name."""
    import models
    assert video_backbone in models.__dict__, 'Unknown model architecture'
    assert audio_backbone in models.__dict__, 'Unknown model architecture'
    video_model = models.__dict__[video_backbone](**video_backbone_args)
    audio_model = models.__dict__[audio_backbone](**audio_backbone_args)
    model = AV_Wrapper(video_model, audio_model, proj_dim=proj_dim)
    if checkpoint is not None:
        ckp = torch.load(checkpoint, map_location='cpu')
        # reference checkpoints carry the DataParallel 'module.' prefix (av_wrapper.py:72-74)
        sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in ckp['model'].items()}
        model.load_state_dict(sd)
    return model
