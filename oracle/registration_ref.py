"""Restatement of the registration path (utils/multiview_registration.py:57-320, 797-1173).

ORACLE / TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.

* warps: the reference's formulas around ``scipy.ndimage`` (the documented semantics of
  ``cupyx.scipy.ndimage``); pinned bit-for-bit by ``tests/golden/registration.npz``
  (reference run with the SciPy stand-in).
* ``phase_cross_correlation``: restates scikit-image's published algorithm (cuCIM mirrors it;
  neither library is installed here) - whole-pixel peak of the phase-normalised
  cross-power spectrum, matrix-multiply upsampled DFT (factor 10 -> 15 x 15 [x 15] samples
  around the peak), and the +-wrap disambiguation by real-space correlation of the 2^ndim
  overlap tiles.  Pinned only by the reference tests' known-answer shifts
  (``tests/test_sofima_deformable_registration.py:121-218``): **parity unpinned** below that.
"""

from __future__ import annotations

import itertools

import numpy as np
import scipy.fft as sfft
import scipy.ndimage as sndi


def overlap_slices_after_translation(shape, translation_px):
    """multiview_registration.py:80-109."""
    slices = []
    for n, t in zip(shape, translation_px, strict=True):
        start = int(np.ceil(max(0.0, -float(t))))
        stop = int(np.floor(min(float(n), float(n) - t)))
        if stop <= start:
            return None
        slices.append(slice(start, stop))
    return tuple(slices)


def pixel_affine(transform_zyx_um, spacing, origin=(0.0, 0.0, 0.0)):
    """matrix_px / offset_px of multiview_registration.py:861-868."""
    spacing = np.asarray(spacing, dtype=np.float32)
    origin = np.asarray(origin, dtype=np.float32)
    T = np.asarray(transform_zyx_um, dtype=np.float32)
    lin, tr = T[:3, :3], T[:3, 3]
    return (lin * spacing[np.newaxis, :]) / spacing[:, np.newaxis], (lin @ origin + tr - origin) / spacing


def warp_array_to_reference(image, transform_zyx_um, spacing, reference_shape, origin=(0, 0, 0),
                            mode="constant", cval=0.0):
    """multiview_registration.py:797-902 with scipy.ndimage.affine_transform(order=1)."""
    m, o = pixel_affine(transform_zyx_um, spacing, origin)
    return sndi.affine_transform(image, matrix=m, offset=o, output_shape=tuple(int(v) for v in reference_shape),
                                 order=1, mode=mode, cval=float(cval))


def _upsampled_dft(data, upsampled_region_size, upsample_factor, axis_offsets):
    """scikit-image ``_upsampled_dft``: one tensordot per axis, last axis first."""
    upsampled_region_size = [upsampled_region_size] * data.ndim
    im2pi = 1j * 2 * np.pi
    for n_items, ups_size, ax_offset in list(zip(data.shape, upsampled_region_size, axis_offsets))[::-1]:
        kernel = (np.arange(ups_size) - ax_offset)[:, None] * sfft.fftfreq(n_items, upsample_factor)
        kernel = np.exp(-im2pi * kernel).astype(data.dtype, copy=False)
        data = np.tensordot(kernel, data, axes=(1, -1))
    return data


def _disambiguate_shift(reference_image, moving_image, shift):
    """scikit-image ``_disambiguate_shift`` (cubic ``ndi.shift`` with grid-wrap when subpixel)."""
    shape = reference_image.shape
    positive_shift = [s % n for s, n in zip(shift, shape)]
    negative_shift = [s - n for s, n in zip(positive_shift, shape)]
    subpixel = np.any(np.array(shift) % 1 != 0)
    shifted = sndi.shift(moving_image, shift, mode="grid-wrap", order=3 if subpixel else 0)
    indices = np.round(positive_shift).astype(int)
    splits = [(slice(0, i), slice(i, None)) for i in indices]
    max_corr, max_slice = -1.0, None
    for test_slice in itertools.product(*splits):
        ref_tile = np.reshape(reference_image[test_slice], -1)
        mov_tile = np.reshape(shifted[test_slice], -1)
        corr = -1.0
        if ref_tile.size > 2:
            with np.errstate(invalid="ignore", divide="ignore"):
                corr = np.corrcoef(ref_tile, mov_tile)[0, 1]
        if corr > max_corr:
            max_corr, max_slice = corr, test_slice
    if max_slice is None:
        return shift
    return np.array([p if sl.stop is None else n
                     for sl, p, n in zip(max_slice, positive_shift, negative_shift)])


def phase_cross_correlation(reference_image, moving_image, upsample_factor=10, disambiguate=True,
                            normalization="phase"):
    """Shift that registers ``moving`` onto ``reference`` (scikit-image semantics, real space,
    normalization="phase" as the reference calls it).  float32 images -> complex64 spectra like the GPU
    libraries.  ``normalization=None, disambiguate=False`` is the scikit-image 0.18 behaviour, which
    ``tests/golden/skimage_phase_correlation.npz`` (real scikit-image 0.18.3) pins."""
    reference_image = np.asarray(reference_image, dtype=np.float32)
    moving_image = np.asarray(moving_image, dtype=np.float32)
    src_freq = sfft.fftn(reference_image)
    target_freq = sfft.fftn(moving_image)
    shape = src_freq.shape
    image_product = src_freq * target_freq.conj()
    if normalization == "phase":
        eps = np.finfo(image_product.real.dtype).eps
        image_product /= np.maximum(np.abs(image_product), 100 * eps)
    elif normalization is not None:
        raise ValueError("normalization must be 'phase' or None")
    cross_correlation = sfft.ifftn(image_product)
    maxima = np.unravel_index(np.argmax(np.abs(cross_correlation)), cross_correlation.shape)
    midpoint = np.array([np.fix(n / 2) for n in shape])
    float_dtype = image_product.real.dtype
    shift = np.stack(maxima).astype(float_dtype, copy=False)
    shift[shift > midpoint] -= np.array(shape)[shift > midpoint]
    if upsample_factor > 1:
        up = np.array(upsample_factor, dtype=float_dtype)
        shift = np.round(shift * up) / up
        region = np.ceil(up * 1.5)
        dftshift = np.fix(region / 2.0)
        offset = dftshift - shift * up
        cc = _upsampled_dft(image_product.conj(), region, up, offset).conj()
        maxima = np.unravel_index(np.argmax(np.abs(cc)), cc.shape)
        maxima = np.stack(maxima).astype(float_dtype, copy=False) - dftshift
        shift = shift + maxima / up
    for dim in range(src_freq.ndim):
        if shape[dim] == 1:
            shift[dim] = 0
    if disambiguate:
        shift = _disambiguate_shift(reference_image, moving_image, shift)
    return np.asarray(shift)


def register_pair_to_fixed(fixed, moving, spacing_zyx_um):
    """Two-stage translation (XY on max-Z projections, then XYZ residual on the overlap)
    (multiview_registration.py:208-320)."""
    if fixed.shape != moving.shape or fixed.ndim != 3:
        raise ValueError("register_pair_to_fixed expects fixed and moving 3D arrays with matching shapes")
    spacing = np.asarray(spacing_zyx_um, dtype=np.float32)
    fp = np.max(np.asarray(fixed, dtype=np.float32), axis=0)
    mp = np.max(np.asarray(moving, dtype=np.float32), axis=0)
    xy_pull = -np.asarray(phase_cross_correlation(fp, mp)).astype(np.float32)
    T = np.eye(4, dtype=np.float32)
    T[1, 3] = float(xy_pull[0]) * float(spacing[1])
    T[2, 3] = float(xy_pull[1]) * float(spacing[2])
    moved = warp_array_to_reference(moving, T, spacing, fixed.shape)
    sl = overlap_slices_after_translation(fixed.shape, (0.0, float(xy_pull[0]), float(xy_pull[1])))
    if sl is None:
        residual_push = np.zeros(3, dtype=np.float32)
    else:
        residual_push = np.asarray(phase_cross_correlation(
            np.asarray(fixed, dtype=np.float32)[sl], np.asarray(moved, dtype=np.float32)[sl])).astype(np.float32)
    total = -residual_push.astype(np.float32)
    total[1] += xy_pull[0]
    total[2] += xy_pull[1]
    out = np.eye(4, dtype=np.float32)
    out[:3, 3] = total * spacing
    return out
