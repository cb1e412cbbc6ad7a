"""Deterministic synthetic Q/K/V generator (counter based, integer exact).

The parity tests, the golden fixtures and ``bench.py`` need the *same* tensors in
this container (where the fixtures are made from the imported reference) and on
the GPU box (where ``/root/reference`` does not exist).  ``torch.randn`` streams
are not guaranteed stable across builds/devices, and libm ``log``/``cos`` can
differ in the last bit between CPUs, so the generator uses integer arithmetic
only:

    z_i = (sum_{k<12} u16(seed, i, k) + 6) / 65536 - 6        (Irwin-Hall, n = 12)

Each ``u16`` is 16 bits of a splitmix64 hash of ``(seed, 3 i + k // 4)``.  The sum
has < 2**20 significant bits, so ``z_i`` is exact in fp32 and its rounding to
bf16/fp16 is unique.  Mean 0, variance 1, support (-6, 6): the "N(0,1) inputs"
of SURVEY.md §8(d), with lighter tails than a true gaussian (excess kurtosis
-0.1) -- stated wherever a number is quoted on it.
"""

from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = x + _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def normal_f32(shape, seed: int, chunk: int = 1 << 22) -> np.ndarray:
    """Return a float32 array of ``shape`` with the integer-exact ~N(0,1) stream ``seed``."""
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.float32)
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([seed], dtype=np.uint64))[0] * np.uint64(3)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        idx = np.arange(lo, hi, dtype=np.uint64) * np.uint64(3)
        acc = np.zeros(hi - lo, dtype=np.int64)
        for w in range(3):
            with np.errstate(over="ignore"):
                h = _splitmix64(idx + np.uint64(w) + base)
            for s in (0, 16, 32, 48):
                acc += ((h >> np.uint64(s)) & np.uint64(0xFFFF)).astype(np.int64)
        # (acc + 6) / 65536 - 6 ; numerator < 2**20 so the fp32 value is exact
        out[lo:hi] = (acc + 6 - 6 * 65536).astype(np.float32) / np.float32(65536.0)
    return out.reshape(shape)


def round_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 bit patterns (uint16), round-to-nearest-even; inputs are finite."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)
    return r.astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << np.uint32(16)).view(np.float32)


def checksum(bits: np.ndarray) -> int:
    """Order-sensitive 64-bit checksum of an integer array (pins generator reproducibility)."""
    a = np.ascontiguousarray(bits).astype(np.uint64).ravel()
    w = (np.arange(a.size, dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
    with np.errstate(over="ignore"):
        return int(np.sum(a * w, dtype=np.uint64))


def qkv(B: int, H: int, Sq: int, Sk: int, D: int, seed: int, dtype: str = "bf16", scale: float = 1.0):
    """Synthetic attention operands in the ``[B, S, H, D]`` layout a fused QKV projection yields.

    Returns torch tensors (q, k, v) of ``dtype`` ("bf16" | "fp16" | "fp32") on the CPU.
    Streams: q -> seed, k -> seed + 1, v -> seed + 2.
    """
    import torch

    td = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[dtype]
    outs = []
    for i, S in enumerate((Sq, Sk, Sk)):
        x = torch.from_numpy(normal_f32((B, S, H, D), seed + i))
        if scale != 1.0:
            x = x * scale
        outs.append(x.to(td))
    return tuple(outs)
