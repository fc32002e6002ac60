"""Deterministic, library-independent tensor generator (test infrastructure).

Every value is a pure function of (name, flat index): a splitmix64 hash of a
64-bit counter, reduced to exact float32 arithmetic (adds/multiplies of small
integers only), so the build container and the GPU box regenerate identical
bits without shipping weights.  Used to make the golden vectors
(oracle/make_golden.py) and to regenerate the same weights/inputs in tests.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform(name: str, shape, lo=-1.0, hi=1.0) -> np.ndarray:
    """float32 uniform in [lo, hi) from 24 hash bits per element."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(_fnv1a64(name))
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + seed
    bits = _splitmix64(ctr) >> np.uint64(40)  # top 24 bits
    u = bits.astype(np.float32) * np.float32(1.0 / 16777216.0)  # exact
    out = u * np.float32(hi - lo) + np.float32(lo)
    return out.reshape(shape)


def normalish(name: str, shape, std=1.0) -> np.ndarray:
    """Irwin-Hall(4) variate scaled to the given std (exact float32 ops)."""
    acc = np.zeros(shape, dtype=np.float32)
    for k in range(4):
        acc = acc + uniform(f"{name}#ih{k}", shape, -1.0, 1.0)
    # var of sum of 4 U(-1,1) = 4/3  ->  scale by sqrt(3/4)
    return acc * np.float32(0.8660254037844386 * std)


def xavier(name: str, shape) -> np.ndarray:
    """U(+-sqrt(6/(fan_in+fan_out))) over the flattened [out, in...] view,
    as nn.init.xavier_uniform_ does (reference model.py:596-603)."""
    fan_out = shape[0]
    fan_in = int(np.prod(shape[1:]))
    a = float(np.sqrt(6.0 / (fan_in + fan_out)))
    return uniform(name, shape, -a, a)
