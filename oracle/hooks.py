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
