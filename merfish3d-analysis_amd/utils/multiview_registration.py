"""Fiducial registration + resampling on MI355X (mirror of utils/multiview_registration.py).

Same function names / arguments / conventions as the reference: images are Z,Y,X; a 4x4
transform is homogeneous in physical Z,Y,X microns and maps *reference/output* coordinates
to *moving/input* coordinates (:814-817).  NumPy arrays in -> NumPy arrays out like the
reference; torch tensors in HBM in -> tensors out with no PCIe traffic.  The device work is
``m3d_max_z``, ``m3d_affine_warp``, ``m3d_flow_warp`` and the phase-correlation kernels of
``libm3d_hip.so``.  The multiview-stitcher adapters of the reference (:130-205, :574-794)
have no hot-path caller and are out of scope.
"""

from __future__ import annotations

import timeit
from collections.abc import Sequence
from typing import Any

import numpy as np

LOCAL_ROUND_TRANSFORM_KEY = "local_round_registered"


def _diag(message: str, *, enabled: bool) -> None:
    if enabled:
        print(f"[multiview-registration] {message}", flush=True)


def _clear_cupy_memory(cp: Any = None) -> None:
    """Release cached device allocations and FFT plans after a registration stage
    (multiview_registration.py:33-54; the name is the reference's - there is no CuPy here: the caches are
    torch's allocator and this package's rocFFT / engine plans)."""
    import gc

    import torch

    from . import rlgc

    if torch.cuda.is_available():
        torch.cuda.synchronize()
    rlgc.clear_rlgc_caches()
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


def _to_device(array, gpu_id: int = 0):
    """(tensor in HBM, was_numpy).  uint16 / float32 keep their dtype, others -> float32."""
    import torch

    if isinstance(array, torch.Tensor):
        t = array if array.is_cuda else array.to(f"cuda:{gpu_id}")
        was_numpy = False
    else:
        arr = np.ascontiguousarray(array)
        if arr.dtype not in (np.uint16, np.float32):
            arr = arr.astype(np.float32)
        t = torch.from_numpy(arr).to(f"cuda:{gpu_id}")
        was_numpy = True
    if t.dtype not in (torch.uint16, torch.float32):
        t = t.to(torch.float32)
    return t.contiguous(), was_numpy


def _max_z_projection_gpu(image, cp: Any = None):
    """Maximum projection over Z as a float32 array in HBM (:57-77)."""
    from .. import kernels

    t, _ = _to_device(image)
    return kernels.max_z_projection(t)


def _overlap_slices_after_translation(shape: Sequence[int], translation_px: Sequence[float]):
    """Output slices whose translated coordinates stay inside the input (:80-109)."""
    slices = []
    for axis_size, t in zip(shape, translation_px, strict=True):
        start = int(np.ceil(max(0.0, -float(t))))
        stop = int(np.floor(min(float(axis_size), float(axis_size) - t)))
        if stop <= start:
            return None
        slices.append(slice(start, stop))
    return tuple(slices)


def _zyx_dict(values: Sequence[float]) -> dict[str, float]:
    values = np.asarray(values, dtype=np.float32).tolist()
    return {"z": float(values[0]), "y": float(values[1]), "x": float(values[2])}


def pixel_affine_from_physical(transform_zyx_um, spacing_zyx_um, origin_zyx_um=(0.0, 0.0, 0.0)):
    """matrix_px, offset_px handed to affine_transform (:861-868), float32 like the reference."""
    spacing = np.asarray(spacing_zyx_um, dtype=np.float32)
    origin = np.asarray(origin_zyx_um, dtype=np.float32)
    transform = np.asarray(transform_zyx_um, dtype=np.float32)
    linear_um = transform[:3, :3]
    translation_um = transform[:3, 3]
    matrix_px = (linear_um * spacing[np.newaxis, :]) / spacing[:, np.newaxis]
    offset_px = (linear_um @ origin + translation_um - origin) / spacing
    return matrix_px, offset_px


def warp_array_to_reference_gpu(
    image,
    *,
    transform_zyx_um: np.ndarray,
    spacing_zyx_um: Sequence[float],
    reference_shape: Sequence[int],
    reference_origin_zyx_um: Sequence[float] = (0.0, 0.0, 0.0),
    mode: str = "constant",
    cval: float = 0.0,
    order: int = 1,
    gpu_id: int = 0,
    z_batch_size: int = 4,
    diagnostics: bool = False,
):
    """Trilinear affine resample into the reference grid (:797-902); dtype follows the input."""
    from .. import kernels

    if order != 1:
        raise NotImplementedError("only order=1 (the hot path's setting) is implemented")
    matrix_px, offset_px = pixel_affine_from_physical(transform_zyx_um, spacing_zyx_um,
                                                      reference_origin_zyx_um)
    start = timeit.default_timer()
    t, was_numpy = _to_device(image, gpu_id)
    if t.dim() != 3:
        raise ValueError(f"Expected a 3D image, got shape {tuple(t.shape)!r}.")
    out = kernels.affine_warp(t, matrix_px, offset_px, reference_shape, mode=mode, cval=cval)
    _diag(f"warp_array_to_reference_gpu_done elapsed_s={timeit.default_timer() - start:.2f}",
          enabled=diagnostics)
    return out.cpu().numpy() if was_numpy else out


def warp_array_to_reference_with_affine_and_sofima_flow_gpu(
    image,
    *,
    transform_zyx_um: np.ndarray,
    spacing_zyx_um: Sequence[float],
    reference_shape: Sequence[int],
    sofima_flow_field_xyz_px,
    flow_field_stride_zyx_px: Sequence[float],
    flow_field_box_start_xyz_px: Sequence[float],
    reference_origin_zyx_um: Sequence[float] = (0.0, 0.0, 0.0),
    mode: str = "constant",
    cval: float = 0.0,
    order: int = 1,
    gpu_id: int = 0,
    z_batch_size: int = 4,
    diagnostics: bool = False,
):
    """Affine composed with a SOFIMA flow field, image sampled once (:905-1131)."""
    import torch

    from .. import kernels

    if order != 1:
        raise NotImplementedError("only order=1 (the hot path's setting) is implemented")
    if len(reference_shape) != 3:
        raise ValueError("reference_shape must have three ZYX elements.")
    t, was_numpy = _to_device(image, gpu_id)
    if t.dim() != 3:
        raise ValueError(f"Expected a 3D image, got shape {tuple(t.shape)!r}.")
    flow = sofima_flow_field_xyz_px
    if not isinstance(flow, torch.Tensor):
        flow = torch.from_numpy(np.ascontiguousarray(flow, dtype=np.float32))
    flow = flow.to(device=t.device, dtype=torch.float32)
    if flow.dim() != 4:
        raise ValueError("sofima_flow_field_xyz_px must have channel plus ZYX axes.")
    if flow.shape[0] != 3 and flow.shape[-1] == 3:
        flow = flow.movedim(-1, 0)
    if flow.shape[0] != 3:
        raise ValueError("SOFIMA flow field must have three XYZ channels.")
    box_start_xyz = np.asarray(flow_field_box_start_xyz_px, dtype=np.float32)
    start = timeit.default_timer()
    out = kernels.flow_warp(
        t, flow.contiguous(), np.asarray(flow_field_stride_zyx_px, dtype=np.float32),
        box_start_xyz[[2, 1, 0]], np.asarray(spacing_zyx_um, dtype=np.float32),
        np.asarray(reference_origin_zyx_um, dtype=np.float32),
        np.asarray(transform_zyx_um, dtype=np.float32), reference_shape, mode=mode, cval=cval)
    _diag("warp_array_to_reference_with_affine_and_sofima_flow_gpu_done "
          f"elapsed_s={timeit.default_timer() - start:.2f}", enabled=diagnostics)
    return out.cpu().numpy() if was_numpy else out


def transform_points_to_reference(
    points_zyx: np.ndarray,
    *,
    transform_zyx_um: np.ndarray,
    spacing_zyx_um: Sequence[float],
    origin_zyx_um: Sequence[float] = (0.0, 0.0, 0.0),
) -> np.ndarray:
    """Native moving-image ZYX pixel points -> reference pixel grid (:1134-1173, host NumPy)."""
    points = np.asarray(points_zyx, dtype=np.float32)
    if points.size == 0:
        return points.reshape((-1, 3))
    spacing = np.asarray(spacing_zyx_um, dtype=np.float32)
    origin = np.asarray(origin_zyx_um, dtype=np.float32)
    physical = points * spacing + origin
    homogeneous = np.concatenate([physical, np.ones((physical.shape[0], 1), dtype=np.float32)], axis=1)
    reference_physical = homogeneous @ np.linalg.inv(transform_zyx_um).T
    return (reference_physical[:, :3] - origin) / spacing


# ----------------------------------------------------------------------------
# phase correlation (cucim.skimage.registration.phase_cross_correlation restated on the device)
# ----------------------------------------------------------------------------
_SPLINE_POLE = np.sqrt(3.0) - 2.0
_PREFILTER_RADIUS = 16  # |pole|^16 = 7e-10: below float32 resolution


def _shift_fir(shift: float, subpixel: bool):
    """FIR taps + integer offset realising ``ndi.shift(..., shift, mode="grid-wrap")`` along one
    axis: cubic B-spline prefilter composed with the 4-tap evaluation (order 3), or a pure
    index shift (order 0)."""
    if not subpixel:
        return np.ones(1, dtype=np.float64), int(np.rint(-shift))
    pos = -float(shift)
    fl = int(np.floor(pos))
    f = pos - fl
    w4 = np.array([(1 - f) ** 3 / 6.0, (3 * f**3 - 6 * f**2 + 4) / 6.0,
                   (-3 * f**3 + 3 * f**2 + 3 * f + 1) / 6.0, f**3 / 6.0])
    k = np.arange(-_PREFILTER_RADIUS, _PREFILTER_RADIUS + 1)
    h = np.sqrt(3.0) * _SPLINE_POLE ** np.abs(k)
    taps = np.convolve(w4, h)  # positions fl-1-R ... fl+2+R
    taps = np.append(taps, 0.0)  # odd length 2 * (R + 2) + 1
    return taps, fl + 1


def _shift_grid_wrap(volume, shift_zyx):
    """``scipy.ndimage.shift(volume, shift, mode="grid-wrap", order=3 if subpixel else 0)``."""
    from .. import kernels

    subpixel = bool(np.any(np.asarray(shift_zyx, dtype=np.float64) % 1 != 0))
    weights, offsets = [], []
    for n, s in zip(volume.shape, shift_zyx):
        if n == 1:
            weights.append(None)
            offsets.append(0)
            continue
        w, off = _shift_fir(float(s), subpixel)
        weights.append(w)
        offsets.append(off)
    return kernels.separable_filter(volume, weights, offsets, wrap=True)


def _disambiguate_shift(reference, moving, shift):
    """Choose, per axis, between the positive and the negative wrap of ``shift`` by the
    real-space correlation of the 2^ndim overlap tiles (scikit-image ``_disambiguate_shift``)."""
    import itertools

    from .. import kernels

    shape = tuple(int(v) for v in reference.shape)
    positive = [float(s) % n for s, n in zip(shift, shape)]
    negative = [p - n for p, n in zip(positive, shape)]
    shifted = _shift_grid_wrap(moving, [float(s) for s in shift])
    indices = [int(i) for i in np.round(positive).astype(int)]
    max_corr, best = -1.0, None
    # choice 0 = slice(0, i) (negative wrap), choice 1 = slice(i, None) (positive shift);
    # itertools.product order and the strict ">" reproduce scikit-image's first-maximum rule
    for combo in itertools.product((0, 1), repeat=3):
        lo = [0 if c == 0 else i for c, i in zip(combo, indices)]
        hi = [i if c == 0 else n for c, i, n in zip(combo, indices, shape)]
        count = int(np.prod([max(h - l, 0) for l, h in zip(lo, hi)]))
        corr = -1.0
        if count > 2:
            n, sa, sb, saa, sbb, sab = kernels.box_corr_sums(reference, shifted, lo, hi)
            va, vb = saa - sa * sa / n, sbb - sb * sb / n
            cov = sab - sa * sb / n
            corr = cov / np.sqrt(va * vb) if va > 0 and vb > 0 else float("nan")
        if corr > max_corr:
            max_corr, best = corr, combo
    if best is None:
        return np.asarray(shift)
    return np.array([p if c == 1 else ng for c, p, ng in zip(best, positive, negative)])


def phase_cross_correlation(reference_image, moving_image, *, upsample_factor: int = 10,
                            disambiguate: bool = True):
    """Translation registering ``moving`` onto ``reference`` (scikit-image / cuCIM semantics:
    real-space inputs, ``normalization="phase"``).  float32 tensors in HBM, 2-D or 3-D.
    Returns a NumPy float array (one entry per axis), like ``phase_cross_correlation(...)[0]``.
    """
    import torch

    from .. import kernels

    ref = reference_image if reference_image.dtype == torch.float32 else reference_image.to(torch.float32)
    mov = moving_image if moving_image.dtype == torch.float32 else moving_image.to(torch.float32)
    if ref.shape != mov.shape:
        raise ValueError("images must be same shape")
    ndim = ref.dim()
    shape = tuple(int(v) for v in ref.shape)
    vol3 = (1,) * (3 - ndim) + shape
    ref3, mov3 = ref.reshape(vol3).contiguous(), mov.reshape(vol3).contiguous()
    plan = kernels.FFTPlan(shape, ref.device)
    try:
        src_freq = plan.execute(kernels.real_to_complex(ref.contiguous()))
        target_freq = plan.execute(kernels.real_to_complex(mov.contiguous()))
        product = kernels.cross_power(src_freq, target_freq, normalize=True, out=src_freq)
        del target_freq
        cc = plan.execute(product.clone(), inverse=True)
        flat = kernels.argmax_abs(cc)
        del cc
    finally:
        plan.close()
    maxima = np.unravel_index(flat, shape)
    f32 = np.float32
    midpoint = np.array([np.fix(n / 2) for n in shape])
    shift = np.stack(maxima).astype(f32)
    over = shift > midpoint
    shift[over] -= np.array(shape, dtype=f32)[over]
    if upsample_factor > 1:
        up = np.array(upsample_factor, dtype=f32)
        shift = np.round(shift * up) / up
        region = int(np.ceil(up * 1.5))
        dftshift = np.fix(region / 2.0)
        offsets = dftshift - shift * up
        data = product
        conj = True  # _upsampled_dft(image_product.conj(), ...)
        for n_items, ax_offset in list(zip(shape, offsets))[::-1]:
            kern = (np.arange(region) - ax_offset)[:, None] * np.fft.fftfreq(n_items, float(up))
            kern = np.exp(-2j * np.pi * kern).astype(np.complex64)
            data = kernels.contract_last(data, torch.from_numpy(kern), conj_data=conj)
            conj = False
        flat = kernels.argmax_abs(data.contiguous())  # |conj(x)| == |x|
        maxima = np.stack(np.unravel_index(flat, tuple(data.shape))).astype(f32) - f32(dftshift)
        shift = shift + maxima / up
    del product
    for dim in range(ndim):
        if shape[dim] == 1:
            shift[dim] = 0
    if disambiguate:
        full = [0.0] * (3 - ndim) + [float(s) for s in shift]
        shift = np.asarray(_disambiguate_shift(ref3, mov3, full))[3 - ndim:]
    return np.asarray(shift)


def register_pair_to_fixed(fixed, moving, *, spacing_zyx_um: Sequence[float], diagnostics: bool = False):
    """Staged phase-correlation translation: XY on max-Z projections, then XYZ on the valid
    overlap of the XY-registered volume (:208-320).  Returns the 4x4 float32 physical
    transform mapping fixed/reference coordinates to moving coordinates."""
    import torch

    from .. import kernels

    if tuple(fixed.shape) != tuple(moving.shape) or fixed.ndim != 3:
        raise ValueError("register_pair_to_fixed expects fixed and moving 3D arrays with "
                         f"matching shapes, got {tuple(fixed.shape)!r} and {tuple(moving.shape)!r}.")
    start = timeit.default_timer()
    spacing = np.asarray(spacing_zyx_um, dtype=np.float32)
    fixed_t, _ = _to_device(fixed)
    moving_t, _ = _to_device(moving)
    xy_push = phase_cross_correlation(kernels.max_z_projection(fixed_t), kernels.max_z_projection(moving_t),
                                      upsample_factor=10, disambiguate=True)
    xy_pull = -np.asarray(xy_push).astype(np.float32)
    xy_transform = np.eye(4, dtype=np.float32)
    xy_transform[1, 3] = float(xy_pull[0]) * float(spacing[1])
    xy_transform[2, 3] = float(xy_pull[1]) * float(spacing[2])
    moved = warp_array_to_reference_gpu(moving_t, transform_zyx_um=xy_transform, spacing_zyx_um=spacing,
                                        reference_shape=tuple(fixed_t.shape), order=1, diagnostics=diagnostics)
    overlap = _overlap_slices_after_translation(tuple(fixed_t.shape),
                                                (0.0, float(xy_pull[0]), float(xy_pull[1])))
    if overlap is None:
        residual_push = np.zeros(3, dtype=np.float32)
    else:
        f = fixed_t[overlap].to(torch.float32).contiguous()
        m = moved[overlap].to(torch.float32).contiguous()
        residual_push = np.asarray(phase_cross_correlation(f, m, upsample_factor=10, disambiguate=True)
                                   ).astype(np.float32)
    total = -residual_push.astype(np.float32, copy=False)
    total[1] += xy_pull[0]
    total[2] += xy_pull[1]
    transform = np.eye(4, dtype=np.float32)
    transform[:3, 3] = total * spacing
    _diag(f"register_pair_to_fixed_done total_pull_shift_px={tuple(float(v) for v in total)} "
          f"elapsed_s={timeit.default_timer() - start:.2f}", enabled=diagnostics)
    return transform
