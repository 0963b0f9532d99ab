"""Host-side helpers for BLS12-381 Fr elements as they cross the C ABI.

An element is a numpy uint64[4]: little-endian limbs of the Montgomery form (R = 2^256), the
in-memory layout of ark_ff::Fp<MontBackend<FrConfig,4>,4>.  These helpers only convert between
that layout and Python integers (for constructing inputs / reading outputs); no table arithmetic
happens here.
"""
from __future__ import annotations

from typing import Iterable, List

import numpy as np

P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
R = (1 << 256) % P
RINV = pow(R, -1, P)
_M64 = (1 << 64) - 1


def from_int(x: int) -> np.ndarray:
    """canonical integer -> Montgomery limbs"""
    m = (int(x) % P) * R % P
    return np.array([(m >> (64 * k)) & _M64 for k in range(4)], dtype=np.uint64)


def to_int(limbs) -> int:
    """Montgomery limbs -> canonical integer"""
    a = np.asarray(limbs, dtype=np.uint64).reshape(4)
    m = int(a[0]) | (int(a[1]) << 64) | (int(a[2]) << 128) | (int(a[3]) << 192)
    if m >= P:
        raise ValueError("non-canonical field element")
    return m * RINV % P


def from_ints(xs: Iterable[int]) -> np.ndarray:
    xs = list(xs)
    out = np.empty((len(xs), 4), dtype=np.uint64)
    for i, x in enumerate(xs):
        out[i] = from_int(x)
    return out


def to_ints(a) -> List[int]:
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return [to_int(row) for row in a]


def add(a, b) -> np.ndarray:
    return from_int(to_int(a) + to_int(b))


ONE = from_int(1)
ZERO = np.zeros(4, dtype=np.uint64)
