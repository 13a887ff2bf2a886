"""NumPy restatement of the reference RLGC deconvolution (utils/rlgc.py).

ORACLE / TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.

Pinned: ``tests/test_oracle_golden.py`` checks this file bit-for-bit against
``tests/golden/rlgc_*.npz``, produced by running the reference's own
``utils/rlgc.py`` with NumPy standing in for CuPy (``rng="numpy"`` mode).
All citations are ``utils/rlgc.py:line`` of QI2lab/merfish3d-analysis v0.13.0.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import scipy.fft as sfft

from . import philox


# ----------------------------------------------------------------------------
# geometry (rlgc.py:72-131, 134-227)
# ----------------------------------------------------------------------------
def next_gpu_fft_size(x: int) -> int:
    """Smallest {2,3}-smooth length >= x (rlgc.py:72-97)."""
    if x <= 1:
        return 1
    n = int(x)
    while True:
        m = n
        while m % 2 == 0:
            m //= 2
        while m % 3 == 0:
            m //= 3
        if m == 1:
            return n
        n += 1


def axis_linear_fft_padding(length: int, psf_support: int, halo_multiplier: int = 1):
    """(before, after) pad widths: PSF half-width halo + FFT smoothing (rlgc.py:100-131)."""
    halo = max((int(psf_support) // 2) * int(halo_multiplier), 0)
    with_halo = length + 2 * halo
    new_length = next_gpu_fft_size(with_halo)
    extra = new_length - with_halo
    return halo + extra // 2, halo + extra - extra // 2


def pad_for_linear_fft(image: np.ndarray, psf_shape, pad_yx: bool = True):
    """np.pad(mode='symmetric') with the widths above (rlgc.py:134-173)."""
    if image.ndim != 3:
        raise ValueError(f"Expected 3D input, got shape {image.shape!r}")
    pad_z = axis_linear_fft_padding(image.shape[0], psf_shape[0])
    if pad_yx:
        pad_y = axis_linear_fft_padding(image.shape[1], psf_shape[1])
        pad_x = axis_linear_fft_padding(image.shape[2], psf_shape[2])
    else:
        pad_y = (0, 0)
        pad_x = (0, 0)
    pad_width = (pad_z, pad_y, pad_x)
    return np.pad(image, pad_width, mode="symmetric"), pad_width


def remove_padding_zyx(padded, pad_width):
    """Crop back to the observed region (rlgc.py:176-200)."""
    sl = []
    for axis, (before, after) in enumerate(pad_width):
        stop = padded.shape[axis] - after if after > 0 else None
        sl.append(slice(before, stop))
    return padded[tuple(sl)]


def symmetric_padded_axis_indices(length: int, pad_before: int, pad_after: int):
    """Source indices of the symmetric extension (rlgc.py:203-227)."""
    observed = np.arange(pad_before, length - pad_after, dtype=np.int64)
    padded = np.pad(observed, (pad_before, pad_after), mode="symmetric")
    return padded[:pad_before], padded[length - pad_after :]


def symmetric_source_index(length: int, pad_before: int, pad_after: int) -> np.ndarray:
    """Full-axis map padded index -> observed source index (identity inside)."""
    observed = np.arange(pad_before, length - pad_after, dtype=np.int64)
    return np.pad(observed, (pad_before, pad_after), mode="symmetric")


def enforce_symmetric_boundary(image: np.ndarray, pad_width) -> None:
    """Axis-sequential overwrite of the pad region, in place (rlgc.py:230-271)."""
    for axis, (before, after) in enumerate(pad_width):
        if before == 0 and after == 0:
            continue
        left, right = symmetric_padded_axis_indices(image.shape[axis], before, after)
        if before > 0:
            dst = [slice(None)] * image.ndim
            dst[axis] = slice(0, before)
            image[tuple(dst)] = np.take(image, left, axis=axis)
        if after > 0:
            dst = [slice(None)] * image.ndim
            dst[axis] = slice(image.shape[axis] - after, None)
            image[tuple(dst)] = np.take(image, right, axis=axis)


def pad_psf(psf_temp: np.ndarray, image_shape, normalize: bool = True) -> np.ndarray:
    """Corner embed, 6 rolls, ifftshift, unit sum (rlgc.py:274-313)."""
    if psf_temp.ndim == 2:
        psf_temp = np.expand_dims(psf_temp, axis=0)
    psf = np.zeros(image_shape, dtype=np.float32)
    psf[: psf_temp.shape[0], : psf_temp.shape[1], : psf_temp.shape[2]] = psf_temp
    for axis, axis_size in enumerate(psf.shape):
        psf = np.roll(psf, int(axis_size / 2), axis=axis)
    for axis, axis_size in enumerate(psf_temp.shape):
        psf = np.roll(psf, -int(axis_size / 2), axis=axis)
    psf = np.fft.ifftshift(psf)
    if normalize:
        s = np.sum(psf)
        psf = psf / (s if s != 0 else 1.0)
    return psf.astype(np.float32)


# 0: numpy.fft, which reproduces the golden vectors (made by the reference running on NumPy) bit for bit.
# > 0: scipy.fft with that many pocketfft worker threads - same mathematics, last-bit differences, and about
# 18x faster on float32 volumes; bench.py's cpu_baseline uses it so that the CPU figure is not handicapped.
FFT_WORKERS = 0


def _rfftn(x):
    return sfft.rfftn(x, workers=FFT_WORKERS) if FFT_WORKERS > 0 else np.fft.rfftn(x)


def _irfftn(x, shape):
    return sfft.irfftn(x, s=shape, axes=tuple(range(-len(shape), 0)), workers=FFT_WORKERS) if FFT_WORKERS > 0 else np.fft.irfftn(x, s=shape, axes=tuple(range(-len(shape), 0)))


def fft_conv(image: np.ndarray, H: np.ndarray, shape) -> np.ndarray:
    """irfftn(rfftn(image) * H, s=shape) through a c64 buffer (rlgc.py:316-350)."""
    z, y, x = shape
    buf = np.empty((z, y, x // 2 + 1), dtype=np.complex64)
    buf[...] = _rfftn(image)
    buf[...] *= H
    return _irfftn(buf, shape).astype(np.float32, copy=False)


def observed_region_mask(shape, pad_width) -> np.ndarray:
    """1 inside the observed image, 0 in padding (rlgc.py:353-379)."""
    mask = np.zeros(shape, dtype=np.float32)
    sl = []
    for axis, (before, after) in enumerate(pad_width):
        stop = shape[axis] - after if after > 0 else None
        sl.append(slice(before, stop))
    mask[tuple(sl)] = 1
    return mask


def kl_div(p: np.ndarray, q: np.ndarray, mask: np.ndarray | None = None) -> float:
    """Masked, self-normalised KL with eps=1e-4 and NaN->0 (rlgc.py:382-412)."""
    eps = np.float32(1e-4)
    if mask is not None:
        p = (p + eps) * mask
        q = (q + eps) * mask
    else:
        p = p + eps
        q = q + eps
    p = p / np.sum(p)
    q = q / np.sum(q)
    with np.errstate(divide="ignore", invalid="ignore"):
        kldiv = p * (np.log(p) - np.log(q))
    kldiv[np.isnan(kldiv)] = 0
    return float(np.sum(kldiv))


# ----------------------------------------------------------------------------
# solver (rlgc.py:497-757)
# ----------------------------------------------------------------------------
@dataclass
class RLGCInfo:
    """Per-call diagnostics the parity tests compare against the HIP path."""

    iterations: int = 0
    stop_reason: str = ""
    padded_shape: tuple = ()
    pad_width: tuple = ()
    kld_image: list = field(default_factory=list)
    kld_split1: list = field(default_factory=list)
    kld_split2: list = field(default_factory=list)
    updated_fraction: list = field(default_factory=list)
    max_relative_delta: list = field(default_factory=list)
    splits: list = field(default_factory=list)


def _draw_split(observed_image, rng_mode, rng, seed, iteration, core_index):
    n = observed_image.astype("int64")
    if rng_mode == "numpy":
        return rng.binomial(n, p=0.5).astype(np.float32)
    if rng_mode == "philox":
        # voxel counter = linear index in the *unpadded* image; pad voxels have n=0
        out = np.zeros(n.shape, dtype=np.float32)
        sel = core_index >= 0
        out[sel] = philox.binomial_half(n[sel], core_index[sel], iteration, seed).astype(
            np.float32
        )
        return out
    raise ValueError(rng_mode)


def rlgc(
    image: np.ndarray,
    psf: np.ndarray,
    safe_mode: bool = True,
    limit: float = 0.01,
    max_delta: float = 0.001,
    pad_yx: bool = True,
    rng_seed: int | None = 42,
    normalize_psf: bool = True,
    rng: str = "numpy",
    injected_splits: list | None = None,
    max_iterations: int | None = None,
    record_splits: bool = False,
    return_info: bool = False,
):
    """Richardson-Lucy gradient-consensus solver, op for op (rlgc.py:497-757).

    ``rng="numpy"`` mirrors the reference run under the NumPy stand-in;
    ``rng="philox"`` uses the counter-based split of the HIP build;
    ``injected_splits`` (list of padded float32 arrays, one per iteration)
    overrides both - the test hook SURVEY.md section 7 asks for.
    ``max_iterations`` is a test-only bound (the reference has none).
    """
    info = RLGCInfo()
    gen = np.random.default_rng(rng_seed) if rng == "numpy" else None
    seed = 0 if rng_seed is None else int(rng_seed)

    if psf.ndim == 2:
        psf = np.expand_dims(psf, axis=0)
    if image.ndim == 2:
        image = np.expand_dims(image, axis=0)

    image_pad_np, pad_width = pad_for_linear_fft(
        image=image, psf_shape=tuple(int(v) for v in psf.shape), pad_yx=pad_yx
    )
    image_gpu = np.asarray(image_pad_np, dtype=np.float32)
    psf_gpu = pad_psf(np.asarray(psf, dtype=np.float32), image_gpu.shape, normalize_psf)

    otf = _rfftn(psf_gpu)  # rlgc.py:582 (NumPy 2 keeps f32 -> c64)
    otfT = np.conjugate(otf)
    otfotfT = otf * otfT

    observed_mask = observed_region_mask(image_gpu.shape, pad_width)
    observed_image = image_gpu * observed_mask
    update_norm = fft_conv(observed_mask, otfT, image_gpu.shape)
    update_norm = np.maximum(update_norm, np.float32(1e-6))

    # linear index of each padded voxel in the unpadded image (-1 in padding)
    core_index = np.full(image_gpu.shape, -1, dtype=np.int64)
    core_view = remove_padding_zyx(core_index, pad_width)
    core_view[...] = np.arange(image.size, dtype=np.int64).reshape(image.shape)

    num_pixels = int(np.sum(observed_mask))
    num_iters = 0
    prev_kld1 = np.inf
    prev_kld2 = np.inf

    recon = np.mean(observed_image[observed_mask > 0]) * np.ones(
        image_gpu.shape, dtype=np.float32
    )
    previous_recon = recon
    info.padded_shape = tuple(int(v) for v in image_gpu.shape)
    info.pad_width = pad_width

    while True:
        if injected_splits is not None:
            split1 = np.asarray(injected_splits[num_iters], dtype=np.float32)
        else:
            split1 = _draw_split(observed_image, rng, gen, seed, num_iters, core_index)
        if record_splits:
            info.splits.append(split1.copy())
        split2 = observed_image - split1

        Hu = fft_conv(recon, otf, image_gpu.shape)

        kldim = kl_div(Hu, observed_image, observed_mask)
        kld1 = kl_div(Hu, split1, observed_mask)
        kld2 = kl_div(Hu, split2, observed_mask)
        info.kld_image.append(kldim)
        info.kld_split1.append(kld1)
        info.kld_split2.append(kld2)

        if safe_mode:
            should_restore = (kld1 > prev_kld1) or (kld2 > prev_kld2)
        else:
            should_restore = (kld1 > prev_kld1) and (kld2 > prev_kld2)
        if should_restore:
            recon[...] = previous_recon
            info.stop_reason = "restore_previous_recon"
            break
        prev_kld1 = kld1
        prev_kld2 = kld2

        with np.errstate(divide="ignore", invalid="ignore"):
            HTratio1 = (
                fft_conv(
                    observed_mask * np.divide(split1, 0.5 * (Hu + 1e-12), dtype=np.float32),
                    otfT,
                    image_gpu.shape,
                )
                / update_norm
            )
            HTratio2 = (
                fft_conv(
                    observed_mask * np.divide(split2, 0.5 * (Hu + 1e-12), dtype=np.float32),
                    otfT,
                    image_gpu.shape,
                )
                / update_norm
            )
        HTratio = HTratio1 + HTratio2

        consensus_map = fft_conv((HTratio1 - 1) * (HTratio2 - 1), otfotfT, recon.shape)

        previous_recon = recon
        # filter_update (rlgc.py:23-31)
        recon = np.where(consensus_map < 0, recon, recon * HTratio).astype(np.float32)
        enforce_symmetric_boundary(recon, pad_width)

        num_updated = np.sum((consensus_map >= 0) * observed_mask)
        observed_recon = recon * observed_mask
        observed_previous = previous_recon * observed_mask
        recon_max = np.maximum(np.max(observed_recon), np.float32(1e-12))
        updated_fraction = float(num_updated) / float(num_pixels)
        max_relative_delta = float(
            np.max(np.abs(observed_recon - observed_previous) / recon_max)
        )
        info.updated_fraction.append(updated_fraction)
        info.max_relative_delta.append(max_relative_delta)

        num_iters += 1
        if updated_fraction < limit:
            info.stop_reason = "limit"
            break
        if max_relative_delta < max_delta:
            info.stop_reason = "max_delta"
            break
        if max_iterations is not None and num_iters >= max_iterations:
            info.stop_reason = "max_iterations"
            break

    info.iterations = num_iters
    out = np.asarray(remove_padding_zyx(recon, pad_width)).astype(np.float32)
    if return_info:
        return out, info
    return out


# ----------------------------------------------------------------------------
# routing (rlgc.py:434-494, 783-1157)
# ----------------------------------------------------------------------------
def resolve_tiled_axis_geometry(requested_crop, image_size, psf_support, axis_name):
    """Retained size + discarded halo (= full PSF support) (rlgc.py:434-467)."""
    if requested_crop <= 0:
        raise ValueError(f"{axis_name} must be greater than 0 for tiled 3D RLGC.")
    retained = min(int(requested_crop), int(image_size))
    if retained >= image_size:
        return retained, 0
    return retained, int(psf_support)


def axis_retained_bounds(retained_size, image_size):
    """Non-overlapping bounds covering [0, image_size) (rlgc.py:470-494)."""
    if retained_size <= 0:
        raise ValueError("retained_size must be greater than 0.")
    bounds, start = [], 0
    while start < image_size:
        stop = min(start + retained_size, image_size)
        bounds.append((start, stop))
        start = stop
    return bounds


def chunked_rlgc(
    image,
    psf,
    crop_yx: int = 2048,
    crop_z=None,
    safe_mode=True,
    limit=0.01,
    max_delta=0.001,
    rng_seed=42,
    normalize_psf=True,
    rng: str = "numpy",
    max_iterations=None,
):
    """Full-frame vs lateral-tile routing without the OOM retry (rlgc.py:783-1157)."""
    if crop_z is not None:
        raise ValueError("RLGC no longer supports axial chunking; leave crop_z=None.")
    if crop_yx <= 0:
        raise ValueError("crop_yx must be greater than 0.")
    image_arr = np.asarray(image)
    if image_arr.ndim == 2:
        work = image_arr[np.newaxis, ...]
    elif image_arr.ndim == 3:
        work = image_arr
    else:
        raise ValueError(f"Expected a 2D or 3D image, got shape {image_arr.shape}")
    psf_arr = np.asarray(psf)
    if psf_arr.ndim not in (2, 3):
        raise ValueError(f"Expected a 2D or 3D PSF, got shape {psf_arr.shape}")
    psf_shape = psf_arr.shape if psf_arr.ndim == 3 else (1, *psf_arr.shape)
    crop_yx = min(int(crop_yx), int(max(work.shape[-2:])))

    kw = dict(
        safe_mode=safe_mode,
        limit=limit,
        max_delta=max_delta,
        normalize_psf=normalize_psf,
        rng=rng,
        max_iterations=max_iterations,
    )
    if crop_yx >= work.shape[-2] and crop_yx >= work.shape[-1]:
        out = rlgc(image_arr, psf, pad_yx=True, rng_seed=rng_seed, **kw)
        if image_arr.ndim == 2 and out.ndim == 3 and out.shape[0] == 1:
            out = np.squeeze(out, axis=0)
        return out

    full = work.shape
    ry, py = resolve_tiled_axis_geometry(crop_yx, full[1], int(psf_shape[-2]), "crop_yx")
    rx, px = resolve_tiled_axis_geometry(crop_yx, full[2], int(psf_shape[-1]), "crop_yx")
    out = np.zeros_like(work, dtype=np.float32)
    bz = axis_retained_bounds(full[0], full[0])
    by = axis_retained_bounds(ry, full[1])
    bx = axis_retained_bounds(rx, full[2])
    tiles = [(z, y, x) for x in bx for y in by for z in bz]  # rlgc.py:928-933
    for tile_idx, ((z0, z1), (y0, y1), (x0, x1)) in enumerate(tiles):
        yc0, yc1 = max(y0 - py, 0), min(y1 + py, full[1])
        xc0, xc1 = max(x0 - px, 0), min(x1 + px, full[2])
        crop = work[z0:z1, yc0:yc1, xc0:xc1]
        dec = rlgc(
            crop,
            psf,
            rng_seed=None if rng_seed is None else rng_seed + tile_idx,
            **kw,
        )
        out[z0:z1, y0:y1, x0:x1] = dec[
            :, y0 - yc0 : y0 - yc0 + (y1 - y0), x0 - xc0 : x0 - xc0 + (x1 - x0)
        ]
    if image_arr.ndim == 2:
        out = np.squeeze(out, axis=0)
    return out
