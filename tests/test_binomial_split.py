"""The bounded-work Binomial(n, 1/2) photon split (csrc/m3d_binom.h, oracle/philox.py; replaces the draw of
utils/rlgc.py:564, 619-622): the library's tables against the oracle's exact integers and against the binomial
distribution itself, the host build of the sampler against the oracle draw by draw, and a chi-square test of every
branch of the sampler (pop-count only, one table row, chunked).  CPU only - the GPU twin of the draw-by-draw
comparison is tests/test_gpu_rlgc.py::test_philox_split_is_bit_identical_to_oracle."""

import ctypes
import math
from fractions import Fraction

import numpy as np
import pytest
from scipy import stats

from conftest import pkg
from oracle import philox


@pytest.fixture(scope="module")
def lib_tables():
    lib = pkg("_hip").lib()
    thr, rows = ctypes.c_void_p(), ctypes.c_void_p()
    n_thr, q_max = ctypes.c_int64(), ctypes.c_int32()
    rc = lib.m3d_binom_tables_host(ctypes.byref(thr), ctypes.byref(n_thr), ctypes.byref(rows), ctypes.byref(q_max))
    assert rc == 0
    t = np.ctypeslib.as_array(ctypes.cast(thr, ctypes.POINTER(ctypes.c_uint64)), shape=(n_thr.value,))
    r = np.ctypeslib.as_array(ctypes.cast(rows, ctypes.POINTER(ctypes.c_uint32)), shape=(q_max.value + 1, 2))
    return t, r, q_max.value


def test_library_tables_equal_the_oracles_exact_integers(lib_tables):
    thr, rows, q_max = lib_tables
    assert q_max == philox.Q_MAX == 1023
    assert rows[0].tolist() == [0, 0] and rows[1].tolist() == [0, 0]
    stride = thr.size // (q_max + 1)
    assert stride * (q_max + 1) == thr.size and stride % 16 == 0
    for q in range(2, q_max + 1):
        off, length = int(rows[q, 0]), int(rows[q, 1])
        ref = philox.binomial_row(q)
        assert length == ref.size and off == q * stride and length + 3 <= stride, q
        assert thr[off] == 0 and np.all(thr[off + 1 + length:off + stride] == np.uint64(2 ** 64 - 1))   # sentinels
        assert np.array_equal(thr[off + 1:off + 1 + length], ref), q


@pytest.mark.parametrize("q", [2, 3, 17])
def test_thresholds_are_the_floored_exact_distribution(q):
    """floor(2^63 P(|X - 32 q| <= i)) with P from exact rational arithmetic."""
    n, m = 64 * q, 32 * q
    row = philox.binomial_row(q)
    acc = Fraction(math.comb(n, m), 2 ** n)
    for i in range(row.size + 1):
        if i:
            acc += Fraction(2 * math.comb(n, m + i), 2 ** n)
        exact = (acc.numerator << 63) // acc.denominator
        if i < row.size:
            assert int(row[i]) == exact, i
        else:
            assert exact >= 2 ** 63 - 1     # the first threshold that is not stored
    assert np.all(np.diff(row.astype(np.float64)) >= 0)


def threshold_probes(rng, q_values, n_random=3000):
    """(q, hi, lo, expected draw): 64-bit uniforms on, just below and just above every threshold of the rows (where a
    search that starts from an approximate quantile must still land on the exact count), the extremes, and random
    ones; expected = 32 q +- #{thresholds <= u >> 1} with the sign from the lowest bit."""
    qs, us = [], []
    for q in q_values:
        T = philox.binomial_row(q).astype(object)
        v = np.concatenate([[t - 1, t, t + 1] for t in T] + [[0, 1, 2 ** 63 - 1, 2 ** 63 - 2]]).astype(object)
        v = np.clip(v, 0, 2 ** 63 - 1)
        u = np.concatenate([v * 2, v * 2 + 1, [int(x) for x in rng.integers(0, 2 ** 64, size=n_random, dtype=np.uint64)]])
        qs.append(np.full(u.size, q))
        us.append(u)
    q = np.concatenate(qs).astype(np.uint32)
    u = np.array([int(x) for x in np.concatenate(us)], dtype=np.uint64)
    hi, lo = (u >> np.uint64(32)).astype(np.uint32), (u & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    want = philox._table_draw(q.astype(np.int64), hi, lo)
    return q, hi, lo, want


def test_host_build_of_the_table_draw_lands_on_the_exact_count(tmp_path):
    import subprocess
    from pathlib import Path

    out = tmp_path / "libemul.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", str(out),
                    str(Path(__file__).resolve().parent / "host_emul" / "host_emul.cpp")], check=True)
    emul = ctypes.CDLL(str(out))
    emul.emul_table_draw.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_void_p]
    q, hi, lo, want = threshold_probes(np.random.default_rng(3), (2, 3, 7, 40, 200, 511, 937, 1023))
    got = np.zeros(q.size, dtype=np.uint32)
    emul.emul_table_draw(q.ctypes.data, hi.ctypes.data, lo.ctypes.data, q.size, got.ctypes.data)
    assert np.array_equal(got.astype(np.int64), want)


def _chi_square(samples, n):
    lo, hi = int(samples.min()), int(samples.max())
    k = np.arange(lo, hi + 1)
    obs = np.bincount(samples - lo, minlength=k.size).astype(np.float64)
    exp = stats.binom.pmf(k, n, 0.5) * samples.size
    # everything outside [lo, hi] belongs to the end cells; merge cells until each expects >= 10 draws
    exp[0] += stats.binom.cdf(lo - 1, n, 0.5) * samples.size
    exp[-1] += stats.binom.sf(hi, n, 0.5) * samples.size
    o, e, co, ce = [], [], 0.0, 0.0
    for a, b in zip(obs, exp):
        co, ce = co + a, ce + b
        if ce >= 10:
            o.append(co)
            e.append(ce)
            co = ce = 0.0
    o[-1] += co
    e[-1] += ce
    o, e = np.asarray(o), np.asarray(e)
    chi2 = float(((o - e) ** 2 / e).sum())
    return chi2, o.size - 1, float(stats.chi2.sf(chi2, o.size - 1))


@pytest.mark.parametrize("n", [37,       # r only: pop-count of 37 bits
                               64,       # q = 1 only: pop-count of words (z, w)
                               101,      # r + q = 1 (the background level of the synthetic tiles)
                               128,      # q = 2: the smallest table row alone
                               1000,     # r = 40, q = 15
                               60000,    # r = 32, q = 937: a saturated camera
                               65535,    # the largest uint16 count, q = 1023
                               200000])  # three chunks of 65472 + remainder (float images only)
def test_draws_follow_the_binomial_distribution(n):
    draws = 300_000
    s = philox.binomial_half(np.full(draws, n), np.arange(draws) + 12345, 5, 0xC0FFEE)
    assert s.min() >= 0 and s.max() <= n
    assert abs(s.mean() - n / 2) < 5 * math.sqrt(n / 4 / draws)
    chi2, dof, p = _chi_square(s, n)
    assert p > 1e-4, (n, chi2, dof, p)


def test_draws_do_not_repeat_across_voxels_iterations_and_seeds():
    n = np.full(50_000, 5000)
    base = philox.binomial_half(n, np.arange(n.size), 1, 7)
    for other in (philox.binomial_half(n, np.arange(n.size) + n.size, 1, 7),
                  philox.binomial_half(n, np.arange(n.size), 2, 7), philox.binomial_half(n, np.arange(n.size), 1, 8)):
        r = np.corrcoef(base, other)[0, 1]
        assert abs(r) < 0.02 and np.mean(base == other) < 0.02
    assert np.array_equal(base, philox.binomial_half(n, np.arange(n.size), 1, 7))
