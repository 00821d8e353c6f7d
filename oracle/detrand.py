"""Deterministic, platform-independent pseudo-random inputs for fixtures and tests.

TEST INFRASTRUCTURE ONLY (see oracle/README.md): counter-based splitmix64 hashing so
the same (seed, shape) yields bit-identical float32 arrays on any numpy version and
on the GPU box, without committing large random inputs.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def bits(seed, n):
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x100000001B3)
        return _splitmix64(_splitmix64(ctr))


def uniform(seed, shape, lo=0.0, hi=1.0):
    """float32 uniform in [lo, hi): 24 random mantissa bits, exact in float32."""
    n = int(np.prod(shape))
    u = (bits(seed, n) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def randint(seed, shape, lo, hi):
    """int64 uniform in [lo, hi)."""
    n = int(np.prod(shape))
    r = (bits(seed, n) >> np.uint64(11)) % np.uint64(hi - lo)
    return (r.astype(np.int64) + lo).reshape(shape)
