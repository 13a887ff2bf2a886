"""Counter-based binomial(n, 1/2) photon split (NumPy side of the HIP sampler).

ORACLE / TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.

The reference draws the RLGC 50:50 split with
``cp.random.default_rng(seed).binomial(n, 0.5)`` (``utils/rlgc.py:564,619-622``),
a cuRAND stream nothing else can reproduce.  The MI355X build replaces it by a
stateless sampler that needs no stored split array: for voxel ``v`` at
iteration ``it`` the split is the pop-count of ``n`` fresh random bits taken
from Philox4x32 blocks (``SPLIT_ROUNDS`` = 7 rounds) with

    key     = (seed & 0xffffffff, seed >> 32)
    counter = (v & 0xffffffff, v >> 32, it, chunk)      chunk = 0, 1, ...

each block supplying 128 bits.  This is an *exact* Binomial(n, 1/2) draw, purely
integer, hence bit-identical between this file and
``merfish3d-analysis_amd/csrc/m3d_philox.h``.
"""

from __future__ import annotations

import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)


SPLIT_ROUNDS = 7  # = kSplitRounds of csrc/m3d_philox.h (BigCrush-clean from 7 rounds on; 10 is Random123's margin)


def philox4x32_10(c0, c1, c2, c3, k0: int, k1: int, rounds: int = 10):
    """Philox4x32 on uint32 counter arrays (Random123 definition; 10 rounds = the published known-answer form)."""
    c0 = np.asarray(c0, dtype=np.uint64)
    c1 = np.asarray(c1, dtype=np.uint64)
    c2 = np.asarray(c2, dtype=np.uint64)
    c3 = np.asarray(c3, dtype=np.uint64)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(int(rounds)):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, lo1, n2, lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return (
        c0.astype(np.uint32),
        c1.astype(np.uint32),
        c2.astype(np.uint32),
        c3.astype(np.uint32),
    )


def _popcount32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    x = x - ((x >> np.uint32(1)) & np.uint32(0x55555555))
    x = (x & np.uint32(0x33333333)) + ((x >> np.uint32(2)) & np.uint32(0x33333333))
    x = (x + (x >> np.uint32(4))) & np.uint32(0x0F0F0F0F)
    return ((x * np.uint32(0x01010101)) >> np.uint32(24)).astype(np.int64)


def binomial_half(n: np.ndarray, voxel_index: np.ndarray, iteration: int, seed: int):
    """Exact Binomial(n, 1/2) per voxel; ``n`` int64 >= 0, ``voxel_index`` int64."""
    n = np.asarray(n, dtype=np.int64).ravel()
    n = np.maximum(n, 0)
    v = np.asarray(voxel_index, dtype=np.uint64).ravel()
    out = np.zeros(n.shape, dtype=np.int64)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    k0, k1 = seed & 0xFFFFFFFF, seed >> 32
    max_chunks = int((n.max() + 127) // 128) if n.size else 0
    for chunk in range(max_chunks):
        remaining = n - 128 * chunk
        sel = np.nonzero(remaining > 0)[0]
        if sel.size == 0:
            break
        vv = v[sel]
        w = philox4x32_10(
            vv & _MASK32,
            vv >> np.uint64(32),
            np.full(sel.shape, iteration, dtype=np.uint64),
            np.full(sel.shape, chunk, dtype=np.uint64),
            k0,
            k1,
            rounds=SPLIT_ROUNDS,
        )
        rem = remaining[sel]
        acc = np.zeros(sel.shape, dtype=np.int64)
        for i in range(4):
            nbits = np.clip(rem - 32 * i, 0, 32)
            mask = np.where(
                nbits >= 32,
                np.uint64(0xFFFFFFFF),
                (np.uint64(1) << nbits.astype(np.uint64)) - np.uint64(1),
            ).astype(np.uint32)
            acc += _popcount32(w[i] & mask)
        out[sel] += acc
    return out
