"""NumPy/SciPy restatement of the PixelDecoder pixel-decoding stages.

ORACLE / TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.

Citations are ``PixelDecoder.py:line`` of QI2lab/merfish3d-analysis v0.13.0.
Pinned by ``tests/golden/decode_pixels_*.npz`` (reference run under NumPy /
SciPy stand-ins) in ``arith="standin"`` mode, which uses exactly the stand-in
arithmetic (BLAS dot, ``np.round`` in float32).  ``arith="device"`` is the
restatement of what the CUDA libraries document and what the HIP kernel
implements bit-for-bit:

* cuVS ``pairwise_distance(metric="euclidean")`` is the *expanded* form
  ``sqrt(max(0, |x|^2 + |y|^2 - 2 x.y))`` in float32; its summation order is
  unspecified, the build fixes it to a k-ordered fmaf chain (the numerics of
  ``v_mfma_f32_32x32x2_f32``) - third-party arithmetic, parity unpinned at the ulp.
* CuPy ``cp.round(x, 5)`` on float32 evaluates ``rint(x * 1e5) / 1e5`` in
  *double* and casts back (``cupy/_core/_routines_math.pyx`` ``_round_float``),
  NumPy does it in float32.
"""

from __future__ import annotations

import numpy as np
import scipy.ndimage as sndi

from . import build as _build


def codebook_thresholds(on_bit_count: int):
    """Pixel / transcript distance thresholds (PixelDecoder.py:778-791)."""
    k = float(on_bit_count)
    pixel = float(np.sqrt(2.0 - 2.0 * ((k - 2.0) / np.sqrt(k * (k - 2.0)))))
    transcript = float(np.sqrt(2.0 - 2.0 * (k / np.sqrt(k * (k + 2.0)))))
    return pixel, transcript


def load_codebook(bits: np.ndarray, gene_ids):
    """Drop one-bit rows, median on-count, thresholds, blanks (PixelDecoder.py:756-800)."""
    bits = np.nan_to_num(np.asarray(bits, dtype=np.float64), nan=0.0)
    on_counts = bits.astype(np.int8).sum(axis=1)
    keep = on_counts != 1
    matrix = bits[keep].astype(int)
    genes = [g for g, k in zip(gene_ids, keep) if k]
    on_bit_count = int(np.median(on_counts[on_counts != 1]))
    pixel_thr, transcript_thr = codebook_thresholds(on_bit_count)
    blank = sum(str(g).lower().startswith("blank") for g in genes)
    return matrix, genes, pixel_thr, transcript_thr, int(blank)


def normalize_codebook(matrix: np.ndarray) -> np.ndarray:
    """Unit-L2 codewords, float64 (PixelDecoder.py:896-906)."""
    m = np.asarray(matrix)
    mag = np.linalg.norm(m, axis=1, keepdims=True)
    mag[mag == 0] = 1
    return m / mag


def lowpass(stack: np.ndarray, sigma, is_3d: bool = True) -> np.ndarray:
    """Per-bit Gaussian, scipy semantics: reflect, truncate 4 (PixelDecoder.py:1948-2024)."""
    if sigma is None or np.any(np.asarray(sigma, dtype=float) == 0):
        return np.asarray(stack, dtype=np.float32)
    out = np.array(stack, dtype=np.float32, copy=True)
    for b in range(out.shape[0]):
        img = np.asarray(stack[b], dtype=np.float32)
        if is_3d:
            out[b] = sndi.gaussian_filter(img, sigma=sigma)
        else:
            for z in range(img.shape[0]):
                out[b, z] = sndi.gaussian_filter(img[z], sigma=(sigma[1], sigma[2]))
    return out


def _round5(x: np.ndarray, arith: str) -> np.ndarray:
    if arith == "standin":
        return np.round(x, 5)
    xd = x.astype(np.float64)
    return (np.rint(xd * 1e5) / 1e5).astype(np.float32)


def distance_argmin(norm_traces_pb: np.ndarray, codebook_f32: np.ndarray, arith: str):
    """min / argmin of the expanded euclidean distance (PixelDecoder.py:2500-2513)."""
    x = np.ascontiguousarray(norm_traces_pb, dtype=np.float32)
    cb = np.ascontiguousarray(codebook_f32, dtype=np.float32)
    if arith == "standin":
        xn = np.sum(x * x, axis=1, dtype=np.float32)[:, None]
        yn = np.sum(cb * cb, axis=1, dtype=np.float32)[None, :]
        d2 = xn + yn - np.float32(2.0) * (x @ cb.T)
        d = np.sqrt(np.maximum(d2, np.float32(0.0))).astype(np.float32)
        return np.min(d, axis=1), np.argmin(d, axis=1)
    dmin = np.empty(x.shape[0], dtype=np.float32)
    imin = np.empty(x.shape[0], dtype=np.int32)
    _build.lib().m3d_oracle_distance_argmin(
        x.ctypes.data, cb.ctypes.data, x.shape[0], x.shape[1], cb.shape[0],
        dmin.ctypes.data, imin.ctypes.data,
    )
    return dmin, imin.astype(np.int64)


def decode_pixels(
    stack_lp: np.ndarray,
    decoding_matrix: np.ndarray,
    pixel_threshold: float,
    magnitude_threshold=(1.1, 2.0),
    background: np.ndarray | None = None,
    normalization: np.ndarray | None = None,
    excluded_indices=(),
    arith: str = "device",
    return_f32: bool = False,
):
    """Per-plane scale -> clip -> L2 normalise -> distance -> call (PixelDecoder.py:2523-2643).

    Returns ``decoded int16[Z,Y,X], magnitude f16, distance f16, scaled f16[B,Z,Y,X]``
    (+ the pre-quantisation float32 magnitude / distance when ``return_f32``).
    """
    B, Z = stack_lp.shape[0], stack_lp.shape[1]
    plane_shape = stack_lp.shape[2:]
    decoded = np.zeros(stack_lp.shape[1:], dtype=np.int16)
    magnitude = np.zeros(stack_lp.shape[1:], dtype=np.float16)
    distance = np.zeros(stack_lp.shape[1:], dtype=np.float16)
    scaled = np.zeros(stack_lp.shape, dtype=np.float16)
    mag32 = np.zeros(stack_lp.shape[1:], dtype=np.float32)
    dist32 = np.zeros(stack_lp.shape[1:], dtype=np.float32)
    cb32 = np.asarray(decoding_matrix).astype(np.float32)
    thr = np.float32(pixel_threshold) if arith == "device" else pixel_threshold
    lo, hi = magnitude_threshold
    for z in range(Z):
        tr = np.asarray(stack_lp[:, z]).reshape(B, -1).astype(np.float32)
        if background is not None and normalization is not None:
            bk = np.asarray(background, dtype=np.float32)[:B]
            nm = np.asarray(normalization, dtype=np.float32)[:B]
            tr = (tr - bk[:, None]) / nm[:, None]  # :2399-2401
        tr = np.clip(tr, 0.0, 1.0)  # :2429 (in place in the reference)
        norms = np.linalg.norm(tr, axis=0)  # :2459
        norms = np.where(norms == 0, np.inf, norms)
        with np.errstate(invalid="ignore"):
            ntr = tr / norms
        norms = np.where(norms == np.inf, -1, norms).astype(np.float32)
        dmin, imin = distance_argmin(ntr.T, cb32, arith)
        dec = np.full(dmin.shape[0], -1, dtype=np.int16)  # :2610
        m = dmin <= thr
        dec[m] = imin[m]
        dec[norms < lo] = -1
        dec[norms > hi] = -1
        if len(excluded_indices):
            dec[np.isin(imin, np.asarray(excluded_indices, dtype=imin.dtype))] = -1  # :863-877
        decoded[z] = dec.reshape(plane_shape)
        magnitude[z] = _round5(norms, arith).reshape(plane_shape)
        scaled[:, z] = _round5(tr, arith).reshape((B, *plane_shape))
        distance[z] = _round5(dmin, arith).reshape(plane_shape)
        mag32[z] = norms.reshape(plane_shape)
        dist32[z] = dmin.reshape(plane_shape)
    if return_f32:
        return decoded, magnitude, distance, scaled, mag32, dist32
    return decoded, magnitude, distance, scaled


def warp_pixel(pixel_zyx, spacing, origin, affine, camera_to_stage=None):
    """Pixel -> physical -> camera/stage -> global affine (PixelDecoder.py:2645-2683)."""
    p = np.asarray(pixel_zyx) * spacing + origin
    if camera_to_stage is not None:
        p = (np.asarray(camera_to_stage) @ np.array([*list(p), 1]))[:-1]
    return (np.array(affine) @ np.array([*list(p), 1]))[:-1]
