"""avid_hip — MI355X-native kernels (libavid_hip.so) + autograd bindings for the AVID/CMA training step.

Importing this package loads the HIP shared library and raises if it is missing: the product path
has no CPU fallback (the CPU restatement lives in ``oracle/`` and is test infrastructure only).
"""
from . import lib  # noqa: F401  (loads libavid_hip.so, raises if absent)
from . import ops  # noqa: F401
from .lib import AvidHipError  # noqa: F401
