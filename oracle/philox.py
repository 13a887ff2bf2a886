"""Counter-based binomial(n, 1/2) photon split (NumPy side of the HIP sampler).

ORACLE / TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.

The reference draws the RLGC 50:50 split with
``cp.random.default_rng(seed).binomial(n, 0.5)`` (``utils/rlgc.py:564,619-622``),
a cuRAND stream nothing else can reproduce.  The MI355X build replaces it by a
stateless sampler that needs no stored split array and ONE Philox4x32 block
(``SPLIT_ROUNDS`` = 7 rounds) per voxel whatever the count:

    key     = (seed & 0xffffffff, seed >> 32)
    counter = (v & 0xffffffff, v >> 32, it, block)      block = 0 (1, 2, ... only above 65535 counts)
    n = 64 q + r
    r < 64  pop-count of r bits of words (x, y)                                  exact
    q = 1   pop-count of words (z, w)                                            exact
    q >= 2  inversion of the 64-bit uniform (z:w): lowest bit = sign, X = 32 q +- D where D is the number of
            thresholds floor(2^63 P(|X - 32 q| <= i)) that are <= the upper 63 bits
    q > 1023: chunks of 64 * 1023, one more 64-bit uniform each ((x:y), (z:w) of blocks 1, 2, ...)

The thresholds are exact integers: ``binomial_row`` evaluates them in 191-bit fixed point with floor divisions,
carries the truncation-error bound and accepts a threshold only when both ends of its error interval floor to the same
integer.  ``merfish3d-analysis_amd/csrc/m3d_binom.h`` certifies its own tables the same way; both are purely integer,
hence bit-identical.
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


Q_MAX = 1023
_P = 191          # fixed-point fraction bits; thresholds = sum >> 128
_V_MAX = (1 << 63) - 1
_central_cache: dict[int, int] = {}
_row_cache: dict[int, np.ndarray] = {}


def _central_terms() -> dict[int, int]:
    """floor-chain of C(2m, m) / 4^m * 2^191 at m = 32 q (truncation error < m units)."""
    if not _central_cache:
        c = 1 << _P
        k = 0
        for q in range(1, Q_MAX + 1):
            m = 32 * q
            while k < m:
                k += 1
                c = c * (2 * k - 1) // (2 * k)
            _central_cache[q] = c
    return _central_cache


def binomial_row(q: int) -> np.ndarray:
    """Thresholds T[i] = floor(2^63 P(|X - 32 q| <= i)), X ~ Binomial(64 q, 1/2), for all i with T[i] < 2^63 - 1."""
    q = int(q)
    if q in _row_cache:
        return _row_cache[q]
    if not 2 <= q <= Q_MAX:
        raise ValueError("rows exist for 2 <= q <= 1023")
    m = 32 * q
    term = total = _central_terms()[q]
    err = m
    rows = []
    i = 0
    while True:
        if i > 0:
            term = term * (m - i + 1) // (m + i)
            total += 2 * term
            err += 2 * (m + i)
        lo = total >> 128
        if lo >= _V_MAX or i >= m:
            break
        if (total + err) >> 128 != lo:
            raise ArithmeticError(f"threshold {i} of row {q} not certified")
        rows.append(lo)
        i += 1
    out = np.asarray(rows, dtype=np.uint64)
    _row_cache[q] = out
    return out


def _table_draw(q: np.ndarray, hi: np.ndarray, lo: np.ndarray) -> np.ndarray:
    """Binomial(64 q, 1/2) for 2 <= q <= Q_MAX from the 64-bit uniforms (hi:lo)."""
    u = (hi.astype(np.uint64) << np.uint64(32)) | lo.astype(np.uint64)
    v = u >> np.uint64(1)
    sign = (u & np.uint64(1)).astype(np.int64)
    out = np.zeros(q.shape, dtype=np.int64)
    for qq in np.unique(q):
        sel = np.nonzero(q == qq)[0]
        d = np.searchsorted(binomial_row(int(qq)), v[sel], side="right").astype(np.int64)
        out[sel] = 32 * int(qq) + np.where(sign[sel] == 1, d, -d)
    return out


def binomial_half(n: np.ndarray, voxel_index: np.ndarray, iteration: int, seed: int):
    """Exact Binomial(n, 1/2) per voxel; ``n`` int64 >= 0, ``voxel_index`` int64."""
    n = np.asarray(n, dtype=np.int64).ravel()
    n = np.maximum(n, 0)
    v = np.asarray(voxel_index, dtype=np.uint64).ravel()
    out = np.zeros(n.shape, dtype=np.int64)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    k0, k1 = seed & 0xFFFFFFFF, seed >> 32
    act = np.nonzero(n > 0)[0]
    if act.size == 0:
        return out

    def block(sel, number):
        vv = v[sel]
        return philox4x32_10(vv & _MASK32, vv >> np.uint64(32), np.full(sel.shape, iteration, dtype=np.uint64),
                             np.full(sel.shape, number, dtype=np.uint64), k0, k1, rounds=SPLIT_ROUNDS)

    def masked_popcount(word, nbits):
        nbits = np.clip(nbits, 0, 32)
        mask = np.where(nbits >= 32, np.uint64(0xFFFFFFFF),
                        (np.uint64(1) << nbits.astype(np.uint64)) - np.uint64(1)).astype(np.uint32)
        return _popcount32(word & mask)

    w = block(act, 0)
    r = n[act] & 63
    acc = masked_popcount(w[0], r) + masked_popcount(w[1], r - 32)
    q = n[act] >> 6
    hi, lo = w[2].copy(), w[3].copy()
    pend_hi, pend_lo = np.zeros_like(hi), np.zeros_like(lo)
    j = 0
    live = np.nonzero(q > 0)[0]           # positions inside `act`
    while live.size:
        qc = np.minimum(q[live], Q_MAX)
        one = qc == 1
        add = np.zeros(live.shape, dtype=np.int64)
        add[one] = _popcount32(hi[live][one]) + _popcount32(lo[live][one])
        if (~one).any():
            add[~one] = _table_draw(qc[~one], hi[live][~one], lo[live][~one])
        acc[live] += add
        q[live] -= qc
        live = live[q[live] > 0]
        if not live.size:
            break
        j += 1                              # uniform j: block (j + 1) // 2, words (x, y) for odd j, (z, w) for even j
        if j & 1:
            w = block(act[live], (j + 1) >> 1)
            hi[live], lo[live] = w[0], w[1]
            pend_hi[live], pend_lo[live] = w[2], w[3]
        else:
            hi[live], lo[live] = pend_hi[live], pend_lo[live]
    out[act] = acc
    return out
