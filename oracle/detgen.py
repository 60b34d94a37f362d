"""Name-keyed deterministic tensor generator (test infrastructure).

The golden fixtures under ``tests/golden/`` were produced by feeding the
reference modules weights and inputs from this generator, so the 85 MB of
R(2+1)D-18 weights never have to be committed: any process that knows the
parameter *name* and *shape* regenerates the identical values with pure integer
numpy arithmetic (no dependence on numpy's / torch's RNG streams).

    value[i] = u24(splitmix64(fnv1a64(name) + i * GOLDEN)) / 2**23 - 1   in [-1, 1)
"""
import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + _GOLDEN) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def det_uniform(name: str, shape) -> np.ndarray:
    """float32 array of ``shape`` with entries in [-1, 1), keyed by ``name``."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        ctr = np.uint64(fnv1a64(name)) + np.arange(n, dtype=np.uint64) * _GOLDEN
    z = splitmix64(ctr)
    u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 23) - 1.0
    return u.astype(np.float32).reshape(shape)


def det_normalish(name: str, shape) -> np.ndarray:
    """Sum of 4 uniforms, variance 1 — a cheap bell-shaped stand-in for randn."""
    acc = np.zeros(shape, dtype=np.float64)
    for k in range(4):
        acc += det_uniform(f"{name}#{k}", shape).astype(np.float64)
    return (acc * np.sqrt(3.0 / 4.0)).astype(np.float32)


def det_indices(name: str, n: int, high: int) -> np.ndarray:
    """int64 indices in [0, high)."""
    with np.errstate(over="ignore"):
        ctr = np.uint64(fnv1a64(name)) + np.arange(n, dtype=np.uint64) * _GOLDEN
    z = splitmix64(ctr)
    return (z % np.uint64(high)).astype(np.int64)


def det_param(name: str, shape) -> np.ndarray:
    """Deterministic value for a model parameter / buffer called ``name``.

    Rules (by suffix of the reference state_dict key):
      * conv / linear ``weight`` (ndim >= 2): uniform * sqrt(6 / fan_in)  (He-uniform)
      * BN ``weight`` (ndim == 1): 1 + 0.25 u ; any ``bias``: 0.1 u
      * ``running_mean``: 0.1 u ; ``running_var``: 1 + 0.25 u ; ``num_batches_tracked``: 0
    """
    shape = tuple(shape)
    if name.endswith("num_batches_tracked"):
        return np.zeros(shape, dtype=np.int64)
    u = det_uniform(name, shape)
    if name.endswith("running_mean"):
        return 0.1 * u
    if name.endswith("running_var"):
        return 1.0 + 0.25 * u
    if name.endswith("bias"):
        return 0.1 * u
    if name.endswith("weight") and len(shape) == 1:
        return 1.0 + 0.25 * u
    fan_in = int(np.prod(shape[1:]))
    return (u * np.sqrt(6.0 / fan_in)).astype(np.float32)
