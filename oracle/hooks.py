"""Test-infrastructure helper: record the device run's fused-ReLU sign patterns so the oracle can be
run with the same pattern (``avid_oracle.RELU_MASKS``).  See the note there for why."""


def capture_relu_masks(model):
    """Forward hooks on every BatchNormCL / hidden LinearCL of ``model`` (the HIP implementation).
    Returns (masks dict keyed + laid out as the oracle expects, remove_fn)."""
    from models.network_blocks import BatchNormCL
    from models.av_wrapper import LinearCL
    masks, handles = {}, []

    def bn_hook(name):
        def f(mod, inp, out):
            t = (out.detach() > 0).permute(0, 4, 1, 2, 3).cpu()
            masks[name] = t[:, :, 0] if name.startswith("audio_model") else t
        return f

    def lin_hook(name):
        def f(mod, inp, out):
            masks[name] = (out.detach() > 0).cpu()
        return f

    for name, mod in model.named_modules():
        if isinstance(mod, BatchNormCL):
            handles.append(mod.register_forward_hook(bn_hook(name)))
        elif isinstance(mod, LinearCL) and not name.endswith(".4"):
            handles.append(mod.register_forward_hook(lin_hook(name)))
    return masks, lambda: [h.remove() for h in handles]


def relu_flip_report(device_masks, oracle_preact):
    """Device ReLU pattern vs the FREE-running oracle's pre-activations, layer by layer.

    Returns (flips, elements, worst) with ``worst`` = the largest |oracle pre-activation| / (layer RMS) over all
    positions whose sign the two implementations disagree on (0 when there is none).  A legitimate disagreement is
    a pre-activation within fp32 summation noise of zero; a wrong mask in a fused epilogue shows up as many flips
    and / or flips at pre-activations of ordinary size."""
    flips = elements = 0
    worst = 0.0
    relu_flip_report.worst_layer = None
    for name, mask in device_masks.items():
        pre = oracle_preact[name]
        assert tuple(pre.shape) == tuple(mask.shape), (name, tuple(pre.shape), tuple(mask.shape))
        diff = mask != (pre > 0)
        n = int(diff.sum())
        flips += n
        elements += mask.numel()
        if n:
            rms = float(pre.double().pow(2).mean().sqrt()) + 1e-30
            w = float(pre[diff].abs().max()) / rms
            if w > worst:
                worst, relu_flip_report.worst_layer = w, (name, n, mask.numel())
    return flips, elements, worst


def capture_pool_argmax(model):
    """Record which position each global max-pool of the device run selected ({"video_model.pool" / "audio_model.pool":
    int64 [B, C] flat (t, h, w) index} — the layout ``avid_oracle.POOL_ARGMAX`` expects).  The selection is read from the
    autograd node of ``ops.global_maxpool`` (its saved argmax tensor: the positions the backward kernel will use).
    Returns (dict, remove_fn)."""
    from avid_hip import ops
    picks, state, handles = {}, {"tower": None}, []
    orig = ops.global_maxpool

    def wrapped(x):
        y = orig(x)
        if state["tower"] is not None and y.grad_fn is not None:
            (am,) = y.grad_fn.saved_tensors
            picks[state["tower"] + ".pool"] = am.detach().long().cpu()
        return y

    def pre(name):
        def f(mod, inp):
            state["tower"] = name
        return f

    for name in ("video_model", "audio_model"):
        handles.append(getattr(model, name).register_forward_pre_hook(pre(name)))
    ops.global_maxpool = wrapped
    # the video stem's fused BatchNorm + ReLU + MaxPool(1,3,3): window slots (dh * 3 + dw) [B,T,Ho,Wo,C] -> [B,C,T,Ho,Wo]
    orig_stem = ops.bn_relu_maxpool

    def wrapped_stem(*a, **k):
        y = orig_stem(*a, **k)
        if y.grad_fn is not None:
            am = y.grad_fn.saved_tensors[3]
            picks["video_model.conv1.pool"] = am.detach().permute(0, 4, 1, 2, 3).contiguous().long().cpu()
        return y

    ops.bn_relu_maxpool = wrapped_stem

    def remove():
        ops.global_maxpool = orig
        ops.bn_relu_maxpool = orig_stem
        for h in handles:
            h.remove()
    return picks, remove


def pool_pick_report(device_picks, oracle_pool_inputs):
    """Device pool selections vs the FREE-running oracle's pool inputs: (disagreements, selections, worst) with
    ``worst`` = the largest (oracle maximum - oracle value at the device's position) / (that tensor's RMS) over the
    disagreeing (sample, channel) pairs — a legitimate disagreement is a near-tie."""
    import torch
    import torch.nn.functional as F
    dis = total = 0
    worst = 0.0
    for key, pick in device_picks.items():
        h = oracle_pool_inputs[key]
        B, C = h.shape[:2]
        if key.endswith("conv1.pool"):             # windowed pool: the oracle's own maxima vs its values at the device's slots
            T, H, W = h.shape[2:]
            Ho, Wo = pick.shape[-2:]
            top = F.max_pool3d(h, (1, 3, 3), (1, 2, 2), (0, 1, 1)).reshape(B, C, -1)
            hh = torch.arange(Ho).view(1, 1, 1, Ho, 1) * 2 - 1 + pick // 3
            ww = torch.arange(Wo).view(1, 1, 1, 1, Wo) * 2 - 1 + pick % 3
            ok = (hh >= 0) & (hh < H) & (ww >= 0) & (ww < W)
            assert bool(ok.all()), "a device pool selection lies in the padding"
            at = h.reshape(B, C, T, H * W).gather(3, (hh * W + ww).reshape(B, C, T, Ho * Wo)).reshape(B, C, -1)
            flat = h.reshape(B, C, -1)
            diff = at != top
            dis += int(diff.sum())
            total += at.numel()
            if diff.any():
                rms = float(flat.double().pow(2).mean().sqrt()) + 1e-30
                worst = max(worst, float((top - at)[diff].max()) / rms)
            continue
        flat = h.reshape(B, C, -1)
        top = flat.max(2).values
        at = flat.gather(2, pick.view(B, C, 1)).squeeze(2)
        # (exact ties — all-zero channels after the ReLU — are not disagreements: the value is the same)
        diff = at != top
        dis += int(diff.sum())
        total += B * C
        if diff.any():
            rms = float(flat.double().pow(2).mean().sqrt()) + 1e-30
            worst = max(worst, float((top - at)[diff].max()) / rms)
    return dis, total, worst
