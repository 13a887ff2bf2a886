"""GPU parity: phase correlation / register_pair_to_fixed against the oracle and the
reference tests' known answers (tests/test_sofima_deformable_registration.py:121-218)."""

import warnings

import numpy as np
import pytest
import torch
from scipy.ndimage import shift as ndi_shift

from conftest import pkg
from oracle import registration_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _blobs(shape, centres, sig):
    z, y, x = np.indices(shape)
    out = np.zeros(shape, dtype=np.float32)
    for z0, y0, x0 in centres:
        out += np.exp(-(((z - z0) / sig[0]) ** 2 + ((y - y0) / sig[1]) ** 2 + ((x - x0) / sig[2]) ** 2))
    return out


def test_spline_shift_grid_wrap_matches_scipy():
    mv = pkg("utils.multiview_registration")
    rng = np.random.default_rng(0)
    vol = rng.normal(size=(9, 40, 37)).astype(np.float32)
    for shift in ((1.3, -2.7, 5.25), (0.0, 12.5, -0.1), (-3.0, 7.0, 2.0), (20.4, -55.1, 40.9)):
        got = mv._shift_grid_wrap(torch.from_numpy(vol).to(DEV), shift).cpu().numpy()
        sub = np.any(np.asarray(shift) % 1 != 0)
        ref = ndi_shift(vol, shift, mode="grid-wrap", order=3 if sub else 0)
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)
    plane = vol[:1]
    got = mv._shift_grid_wrap(torch.from_numpy(plane).to(DEV), (0.0, 1.5, -2.25)).cpu().numpy()
    np.testing.assert_allclose(got[0], ndi_shift(plane[0], (1.5, -2.25), mode="grid-wrap", order=3), atol=2e-5)


@pytest.mark.parametrize("shape,true_shift", [((31, 64, 64), (2.0, 0.0, 0.0)), ((24, 48, 48), (1.0, 8.0, -7.0)),
                                              ((16, 50, 45), (-1.4, 3.3, 6.7))])
def test_phase_cross_correlation_matches_oracle(shape, true_shift):
    mv = pkg("utils.multiview_registration")
    fixed = _blobs(shape, [(shape[0] * 0.3, shape[1] * 0.3, shape[2] * 0.4), (shape[0] * 0.6, shape[1] * 0.7, shape[2] * 0.5),
                           (shape[0] * 0.75, shape[1] * 0.45, shape[2] * 0.8)], (1.9, 3.5, 3.5))
    moving = ndi_shift(fixed, shift=true_shift, order=1, mode="constant", cval=0.0).astype(np.float32)
    ref = R.phase_cross_correlation(fixed, moving)
    got = mv.phase_cross_correlation(torch.from_numpy(fixed).to(DEV), torch.from_numpy(moving).to(DEV))
    np.testing.assert_allclose(got, ref, atol=1e-3)
    np.testing.assert_allclose(-np.asarray(got), true_shift, atol=0.35)
    # 2-D (max projections)
    ref2 = R.phase_cross_correlation(fixed.max(axis=0), moving.max(axis=0))
    got2 = mv.phase_cross_correlation(torch.from_numpy(fixed.max(axis=0)).to(DEV),
                                      torch.from_numpy(moving.max(axis=0)).to(DEV))
    np.testing.assert_allclose(got2, ref2, atol=1e-3)


def test_register_pair_to_fixed_recovers_z_shift_for_warp_contract():
    """Mirror of the reference test (:121-172): z shift +-0.25 px, warp RMSE < 1e-3."""
    mv = pkg("utils.multiview_registration")
    fixed = _blobs((31, 64, 64), [(8, 18, 20), (16, 42, 35), (23, 30, 50)], (2.0, 4.0, 4.0))
    moving = ndi_shift(fixed, shift=(2.0, 0.0, 0.0), order=1, mode="constant", cval=0.0).astype(np.float32)
    spacing = (0.32, 0.098, 0.098)
    T = mv.register_pair_to_fixed(fixed, moving, spacing_zyx_um=spacing)
    assert T.dtype == np.float32 and T.shape == (4, 4)
    np.testing.assert_allclose(np.asarray(T[:3, 3]) / np.asarray(spacing), (2.0, 0.0, 0.0), atol=0.25)
    np.testing.assert_allclose(T, R.register_pair_to_fixed(fixed, moving, spacing), atol=1e-4)
    warped = mv.warp_array_to_reference_gpu(moving, transform_zyx_um=T, spacing_zyx_um=spacing,
                                            reference_shape=fixed.shape, gpu_id=0)
    assert np.sqrt(np.mean((warped - fixed) ** 2)) < 1e-3


def test_register_pair_to_fixed_residual_uses_valid_lateral_overlap():
    """Mirror of the reference test (:175-218): (1, 8, -7) px +-0.35, no warnings."""
    mv = pkg("utils.multiview_registration")
    fixed = _blobs((24, 48, 48), [(7, 12, 14), (12, 30, 27), (18, 22, 38)], (1.8, 3.2, 3.2))
    moving = ndi_shift(fixed, shift=(1.0, 8.0, -7.0), order=1, mode="constant", cval=0.0).astype(np.float32)
    spacing = (0.32, 0.098, 0.098)
    with warnings.catch_warnings():
        warnings.simplefilter("error", UserWarning)
        T = mv.register_pair_to_fixed(fixed, moving, spacing_zyx_um=spacing)
    np.testing.assert_allclose(np.asarray(T[:3, 3]) / np.asarray(spacing), (1.0, 8.0, -7.0), atol=0.35)
    # uint16 inputs, tensors in HBM
    fu = torch.from_numpy((fixed * 20000).astype(np.uint16)).to(DEV)
    mu = torch.from_numpy((moving * 20000).astype(np.uint16)).to(DEV)
    Tu = mv.register_pair_to_fixed(fu, mu, spacing_zyx_um=spacing)
    np.testing.assert_allclose(np.asarray(Tu[:3, 3]) / np.asarray(spacing), (1.0, 8.0, -7.0), atol=0.35)
    with pytest.raises(ValueError):
        mv.register_pair_to_fixed(fixed, moving[:, :10], spacing_zyx_um=spacing)


def test_phase_cross_correlation_on_the_scikit_image_fixture(golden):
    """The images of tests/golden/skimage_phase_correlation.npz (real scikit-image 0.18.3 answers):
    the HIP path equals the oracle's phase-normalised variant and stays within one upsampling step
    of scikit-image's un-normalised answer."""
    mv = pkg("utils.multiview_registration")
    s = golden("skimage_phase_correlation")
    for i in range(4):
        ref, mov = s[f"ref_{i}"], s[f"mov_{i}"]
        got = mv.phase_cross_correlation(torch.from_numpy(ref).to(DEV), torch.from_numpy(mov).to(DEV),
                                         upsample_factor=10, disambiguate=False)
        want = R.phase_cross_correlation(ref, mov, upsample_factor=10, disambiguate=False)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-5)
        assert np.abs(got - s[f"shift_{i}_up10"]).max() <= 0.1 + 1e-5
