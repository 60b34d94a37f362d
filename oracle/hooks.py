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
    for name, mask in device_masks.items():
        pre = oracle_preact[name]
        assert tuple(pre.shape) == tuple(mask.shape), (name, tuple(pre.shape), tuple(mask.shape))
        diff = mask != (pre > 0)
        n = int(diff.sum())
        flips += n
        elements += mask.numel()
        if n:
            rms = float(pre.double().pow(2).mean().sqrt()) + 1e-30
            worst = max(worst, float(pre[diff].abs().max()) / rms)
    return flips, elements, worst
