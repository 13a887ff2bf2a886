"""GPU: the packed complex arithmetic of the FFT passes (csrc/m3d_fft.h), one instruction form at a time.

The products with +-i, the complex products and the Hermitian split / merge terms are written as single
``v_pk_*_f32`` instructions with explicit ``op_sel`` / ``neg`` modifiers (the compiler does not select them), so every
form is pinned here against the arithmetic it stands for, and the 3- / 4- / 8- / 9- / 27-point butterflies built from them
against ``numpy.fft`` - the transform the reference's convolutions use (rlgc.py:424-436 via cupy.fft / numpy.fft)."""

import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu


def _run(n=4096, seed=3):
    import torch
    kernels = pkg("kernels")
    rng = np.random.default_rng(seed)
    a, b, c = ((rng.normal(size=n) + 1j * rng.normal(size=n)).astype(np.complex64) for _ in range(3))
    c[:] = c[0].real + 1j * c.imag          # slot 5 / 6 take the scalar from a wave-uniform register: same real part
    dev = torch.device("cuda:0")
    out = kernels.fft_debug_pk(*(torch.from_numpy(x).to(dev) for x in (a, b, c))).cpu().numpy()
    return a.astype(np.complex128), b.astype(np.complex128), c.astype(np.complex128), out


def test_single_instruction_forms():
    a, b, c, out = _run()
    s = c.real
    want = {
        0: a + 1j * b,
        1: a - 1j * b,
        2: np.conj(a) + 1j * np.conj(b),
        3: a + np.conj(b),
        4: -1j * (a - np.conj(b)),
        5: b + 1j * s * a,
        6: b - 1j * s * a,
        7: a * b,
        8: a * np.conj(b),
        9: a * (0.6 - 0.8j),                 # ctw<forward>: exp(-i theta) with (cos, sin) = (0.6, 0.8)
    }
    for slot, w in want.items():
        np.testing.assert_allclose(out[:, slot], w, rtol=0, atol=2e-6 * max(1.0, np.abs(w).max()), err_msg=f"slot {slot}")
    # the one-rounding forms are exact in float32
    for slot in (0, 1, 2, 3, 4):
        w = want[slot]
        exact = (w.real.astype(np.float32) + 1j * w.imag.astype(np.float32)).astype(np.complex64)
        np.testing.assert_array_equal(out[:, slot], exact, err_msg=f"slot {slot}")


def test_butterflies_are_dfts():
    a, b, c, out = _run(seed=5)
    seq = np.stack([a, b, c, a + b, b - c, c.real * a.real + 1j * c.imag * a.imag, a - c, b + b, c - a], axis=1)
    at = 10
    for radix in (3, 4, 8, 9):
        for inverse in (False, True):
            x = seq[:, :radix]
            full = np.fft.ifft(x, axis=1) * radix if inverse else np.fft.fft(x, axis=1)
            got = out[:, at:at + radix]
            at += radix
            scale = np.abs(full).max()       # outputs in natural order for every radix
            np.testing.assert_allclose(got, full, rtol=0, atol=4e-6 * scale, err_msg=f"radix {radix} inverse {inverse}")
    # radix 27 - the twiddle-free last forward / first inverse stage of the 9 * 9 * 27 Y passes of the headline length
    x27 = np.concatenate([seq, seq * b[:, None], seq * c[:, None]], axis=1)
    for inverse in (False, True):
        full = np.fft.ifft(x27, axis=1) * 27 if inverse else np.fft.fft(x27, axis=1)
        got = out[:, at:at + 27]
        at += 27
        np.testing.assert_allclose(got, full, rtol=0, atol=6e-6 * np.abs(full).max(), err_msg=f"radix 27 inverse {inverse}")
    assert at == out.shape[1]
