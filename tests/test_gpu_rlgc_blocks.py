"""GPU parity: the solver's building blocks as stand-alone functions of utils.rlgc (the reference's module-level
names: filter_update, enforce_symmetric_boundary, pad_psf, fft_conv, _observed_region_mask, kl_div) against the
NumPy restatement the bit-exact RLGC goldens pin (oracle/rlgc_ref.py)."""

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import rlgc_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _psf(shape, sig=(1.6, 1.4, 1.4)):
    z, y, x = np.indices(shape, dtype=np.float64) - (np.asarray(shape)[:, None, None, None] // 2)
    p = np.exp(-0.5 * ((z / sig[0]) ** 2 + (y / sig[1]) ** 2 + (x / sig[2]) ** 2))
    return (p / p.sum() * 1.7).astype(np.float32)   # not unit sum: normalize=True has something to do


def test_filter_update_is_the_reference_elementwise_kernel():
    rl = pkg("utils.rlgc")
    rng = np.random.default_rng(0)
    recon = rng.uniform(0, 50, size=(5, 33, 41)).astype(np.float32)
    ht = rng.uniform(0.5, 1.5, size=recon.shape).astype(np.float32)
    cons = rng.normal(size=recon.shape).astype(np.float32)
    cons[0, 0, :4] = [0.0, -0.0, np.nan, -1e-30]
    want = np.where(cons < 0, recon, recon * ht)
    got = rl.filter_update(_t(recon), _t(ht), _t(cons))
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(rl.filter_update(recon, ht, cons), want)          # NumPy in -> NumPy out
    with pytest.raises(ValueError):
        rl.filter_update(_t(recon), _t(ht[:2]), _t(cons))


@pytest.mark.parametrize("shape,pads", [((12, 30, 40), ((3, 2), (5, 6), (7, 4))), ((4, 9, 10), ((0, 0), (0, 3), (2, 0))),
                                         ((12, 20, 18), ((5, 5), (9, 9), (8, 8)))])
def test_symmetric_boundary_and_mask_match_the_restatement(shape, pads):
    rl = pkg("utils.rlgc")
    rng = np.random.default_rng(1)
    vol = rng.normal(size=shape).astype(np.float32)
    want = vol.copy()
    rlgc_ref.enforce_symmetric_boundary(want, pads)
    dev = _t(vol)
    rl.enforce_symmetric_boundary(dev, pads)
    assert np.array_equal(dev.cpu().numpy(), want)
    assert np.array_equal(rl._observed_region_mask(shape, pads).cpu().numpy(), rlgc_ref.observed_region_mask(shape, pads))
    for length, (a, b) in zip(shape, pads):
        left, right = rl._symmetric_padded_axis_indices(length, a, b)
        rl_l, rl_r = rlgc_ref.symmetric_padded_axis_indices(length, a, b)
        assert np.array_equal(left, rl_l) and np.array_equal(right, rl_r)
    with pytest.raises(ValueError):
        rl.enforce_symmetric_boundary(dev, ((shape[0], 1), (0, 0), (0, 0)))  # no observed region left
    with pytest.raises(ValueError):
        rl.enforce_symmetric_boundary(vol, pads)                          # in place: needs a tensor in HBM


@pytest.mark.parametrize("psf_shape,shape,normalize", [((5, 7, 7), (16, 36, 48), True), ((4, 6, 5), (8, 27, 32), True),
                                                        ((5, 7, 7), (16, 36, 48), False), ((7, 7), (1, 16, 18), True)])
def test_pad_psf_places_every_sample_where_the_reference_does(psf_shape, shape, normalize):
    rl = pkg("utils.rlgc")
    psf = _psf(psf_shape if len(psf_shape) == 3 else (1, *psf_shape)).reshape(psf_shape)
    want = rlgc_ref.pad_psf(psf, shape, normalize)
    got = rl.pad_psf(_t(psf), shape, normalize).cpu().numpy()
    assert got.shape == tuple(shape) and np.array_equal(got != 0, want != 0)
    np.testing.assert_allclose(got, want, rtol=2e-7, atol=0)                # sum order of the normaliser
    if not normalize:
        assert np.array_equal(got, want)


@pytest.mark.parametrize("shape", [(16, 36, 48), (8, 27, 32), (32, 81, 64), (64, 36, 48), (128, 27, 32), (256, 27, 32), (48, 32, 36),
                                   # Y lengths with static Y kernels of their own (radix 8 / 4 stages)
                                   (32, 2048, 36), (16, 1728, 27), (16, 1296, 32), (16, 1152, 36),
                                   # X lengths with radix-8 register edge stages
                                   (16, 27, 128), (16, 32, 192), (8, 36, 576), (8, 27, 1536), (8, 32, 2048),
                                   # 2592: radix-9 edges with 288 edge items (X), column tiles of two (Y: 83 KB at four)
                                   (8, 27, 2592), (8, 2592, 36), (8, 27, 648), (16, 576, 36),
                                   # the headline lateral geometry (BASELINE configs[1] pads 2048 to 2187 = 9 9 9 3):
                                   # static X and Y kernels together, each axis alone with the run-time partner
                                   (8, 2187, 2187), (8, 27, 2187), (8, 2187, 36)])
def test_fft_conv_with_transfer_function_views(shape, monkeypatch):
    rl = pkg("utils.rlgc")
    if int(np.prod(shape)) > 10_000_000:   # pocketfft worker threads for the oracle's transforms at 38 M voxels
        import os

        monkeypatch.setattr(rlgc_ref, "FFT_WORKERS", min(16, os.cpu_count() or 1))
    rng = np.random.default_rng(2)
    img = rng.uniform(0, 100, size=shape).astype(np.float32)
    psf_pad = rlgc_ref.pad_psf(_psf((5, 7, 7)), shape, True)
    otf = np.fft.rfftn(psf_pad)
    H = rl.transfer_function(_t(psf_pad))
    assert H.shape == tuple(shape)
    for view, ref_h in ((H, otf), (H.conj(), np.conjugate(otf)), (H.abs2(), otf * np.conjugate(otf)),
                        (H.conj().conj(), otf)):
        want = rlgc_ref.fft_conv(img, ref_h.astype(np.complex64), shape)
        got = rl.fft_conv(_t(img), view, shape).cpu().numpy()
        assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max()
    assert np.abs(rl.fft_conv(img, H) - rlgc_ref.fft_conv(img, otf.astype(np.complex64), shape)).max() <= 2e-2
    assert (0, *shape) in rl._fft_cache_3d
    with pytest.raises(TypeError):
        rl.fft_conv(_t(img), _t(psf_pad), shape)
    with pytest.raises(ValueError):
        rl.fft_conv(_t(img[:, :, :-2].copy()), H)
    rl.clear_rlgc_caches()
    assert not rl._fft_cache_3d
    with pytest.raises(NotImplementedError):
        rl.transfer_function(_t(np.zeros((8, 10, 14), dtype=np.float32)))   # 10 and 14 are not 2^a 3^b


def test_kl_div_masked_and_plain():
    rl = pkg("utils.rlgc")
    rng = np.random.default_rng(3)
    shape, pads = (10, 40, 50), ((2, 2), (5, 5), (6, 4))
    p = rng.uniform(0, 200, size=shape).astype(np.float32)
    q = (p * rng.uniform(0.8, 1.25, size=shape)).astype(np.float32)
    q[3, 3, :5] = 0.0
    mask = rlgc_ref.observed_region_mask(shape, pads)
    for m in (None, mask):
        want = rlgc_ref.kl_div(p, q, m)
        got = rl.kl_div(_t(p), _t(q), None if m is None else _t(m))
        assert isinstance(got, float) and abs(got - want) <= 2e-6 + 1e-4 * abs(want), (got, want)
    assert rl.kl_div(_t(p), _t(p)) == pytest.approx(0.0, abs=1e-9)
    assert abs(rl.kl_div(p, q, mask) - rlgc_ref.kl_div(p, q, mask)) <= 2e-6 + 1e-4 * abs(rlgc_ref.kl_div(p, q, mask))
