"""``utils.main_utils`` for a run of the reference's ``main-avid.py`` on this package: the reference's OWN module, executed
unmodified from wherever it lies on ``sys.path``, with two names resolved differently.

``utils/main_utils.py:112`` builds ``torch.nn.parallel.DistributedDataParallel(model, device_ids=[gpu])`` and ``:250``
``torch.optim.Adam(params, lr, weight_decay, betas)``.  Both work on this package's modules as they are, at
0.79 of the step engine's throughput: torch's reducer copies every gradient into and out of its buckets and broadcasts 126
buffers one by one, its Adam walks 141 tensors.  ``avid_hip.parallel.DistributedDataParallel`` / ``Adam`` are the same two objects
for that loop (same constructors, same behaviour towards ``main-avid.py:155-180``; flat buffers underneath: 0.93) — INTEGRATION.md 3
asked the user to swap the two factory lines.  This module does it without touching the reference: it finds the NEXT
``main_utils.py`` along ``utils.__path__`` (the reference checkout's), executes that file's code in this module's namespace, and
then binds the module-level name ``torch`` — through which the reference's functions reach both factories at call time — to a
proxy that forwards everything to torch except ``nn.parallel.DistributedDataParallel`` and ``optim.Adam`` — the latter only for a
parameter list that IS the wrapped model's flat-buffer set (main-avid.py's case); any other list (the evaluation scripts' bare models
and parameter subsets) gets torch's own Adam.

Nothing of the reference is copied or edited; without a reference checkout on the path the import fails as it would have.
``AVID_DROPIN=0`` leaves ``torch`` alone (the reference's module as it is)."""
import os as _os
import sys as _sys


def _reference_file():
    import utils as _pkg
    here = _os.path.dirname(_os.path.abspath(__file__))
    for d in list(_pkg.__path__) + [_os.path.join(p, "utils") for p in _sys.path if p]:
        f = _os.path.join(d, "main_utils.py")
        if _os.path.isfile(f) and _os.path.abspath(d) != here:
            return f
    raise ImportError("utils.main_utils: no main_utils.py of the reference checkout on sys.path (put the AVID-CMA directory "
                      "behind avid-cma_amd on PYTHONPATH)")


class _Forward:
    """Attribute access forwarded to ``target`` except for the names in ``over``."""

    def __init__(self, target, over):
        object.__setattr__(self, "_target", target)
        object.__setattr__(self, "_over", over)

    def __getattr__(self, name):
        over = object.__getattribute__(self, "_over")
        if name in over:
            return over[name]
        return getattr(object.__getattribute__(self, "_target"), name)

    def __setattr__(self, name, value):
        setattr(object.__getattribute__(self, "_target"), name, value)

    def __dir__(self):
        return dir(object.__getattribute__(self, "_target"))


def _adam_factory():
    """`torch.optim.Adam` as `build_optimizer` sees it: the flat-buffer Adam exactly where it is the step engine's — the parameters
    are the set a live `avid_hip.parallel.DistributedDataParallel` wrapper (or TrainStep) holds in its flat buffers, which is what
    main-avid.py:93-108 produces — and torch's own Adam for everything else the reference builds optimizers for (the evaluation
    scripts: bare models, a classifier's parameter subset, several groups), whose semantics towards parameters without a gradient
    the one-launch step does not share."""
    import torch
    from avid_hip import parallel

    def Adam(params, *args, **kwargs):
        ps = list(params)
        plain = all(isinstance(p, torch.Tensor) for p in ps)
        if plain and ps and parallel.flat_of([p for p in ps if p.requires_grad]) is not None:
            try:
                return parallel.Adam(ps, *args, **kwargs)
            except (NotImplementedError, ValueError):
                pass
        return torch.optim.Adam(ps, *args, **kwargs)
    Adam.__doc__ = parallel.Adam.__doc__
    return Adam


def _ddp_factory():
    """`torch.nn.parallel.DistributedDataParallel` as `distribute_model_to_cuda` sees it: this build's wrapper for this build's two-tower
    model (`models.av_wrapper`'s class), torch's own for any other module or for arguments the wrapper does not reproduce."""
    import torch
    from avid_hip import parallel

    class DistributedDataParallel(parallel.DistributedDataParallel):
        def __new__(cls, module, *args, **kwargs):
            from models.av_wrapper import AV_Wrapper
            if not isinstance(module, AV_Wrapper):
                return torch.nn.parallel.DistributedDataParallel(module, *args, **kwargs)
            return super().__new__(cls)
    DistributedDataParallel.__name__ = DistributedDataParallel.__qualname__ = "DistributedDataParallel"
    return DistributedDataParallel


def _torch_with_dropins():
    import torch
    from avid_hip import parallel
    nn_parallel = _Forward(torch.nn.parallel, {"DistributedDataParallel": _ddp_factory()})
    nn = _Forward(torch.nn, {"parallel": nn_parallel})
    optim = _Forward(torch.optim, {"Adam": _adam_factory()})
    return _Forward(torch, {"nn": nn, "optim": optim})


REFERENCE_FILE = _reference_file()
with open(REFERENCE_FILE, "rb") as _f:
    exec(compile(_f.read(), REFERENCE_FILE, "exec"), globals())      # the reference's module body, in this namespace
DROPIN = _os.environ.get("AVID_DROPIN", "1") != "0"
if DROPIN:
    torch = _torch_with_dropins()       # what distribute_model_to_cuda / build_optimizer look up when they are CALLED
