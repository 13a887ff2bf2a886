"""Thin torch-tensor wrappers over the C ABI (one function per ``m3d_*`` entry point).

These are the only callers of ``libm3d_hip.so`` besides ``utils/rlgc.py``; the
operator classes (``PixelDecoder``, ``DataRegistration``) and the L3 utility
functions are written on top of them.  Every function takes tensors that
already live in HBM and enqueues on the current torch stream.
"""

from __future__ import annotations

import numpy as np
import torch

from . import _hip


def _stream(t: torch.Tensor) -> int:
    return _hip.current_stream(t.device)


def clip_to_u16(volume: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """``np.clip(x, 0, 65535).astype(np.uint16)`` (DataRegistration.py:435, 993-995)."""
    volume = volume.contiguous()
    if volume.dtype != torch.float32:
        raise ValueError("clip_to_u16 expects float32")
    if out is None:
        out = torch.empty(volume.shape, dtype=torch.uint16, device=volume.device)
    with torch.cuda.device(volume.device):
        _hip.check(_hip.lib().m3d_clip_to_u16(volume.data_ptr(), out.data_ptr(), volume.numel(),
                                              _stream(volume)), "m3d_clip_to_u16")
    return out


def weight_readout(readout_u16: torch.Tensor, predictor_f32: torch.Tensor,
                   out: torch.Tensor | None = None) -> torch.Tensor:
    """``float32(readout) * float32(predictor)`` (PixelDecoder.py:1879-1881)."""
    if readout_u16.element_size() != 2 or predictor_f32.dtype != torch.float32:
        raise ValueError("weight_readout expects uint16 readout and float32 predictor")
    if readout_u16.shape != predictor_f32.shape:
        raise ValueError("readout and predictor shapes differ")
    readout_u16 = readout_u16.contiguous()
    predictor_f32 = predictor_f32.contiguous()
    if out is None:
        out = torch.empty(readout_u16.shape, dtype=torch.float32, device=readout_u16.device)
    with torch.cuda.device(out.device):
        _hip.check(_hip.lib().m3d_weight_readout(readout_u16.data_ptr(), predictor_f32.data_ptr(),
                                                 out.data_ptr(), out.numel(), _stream(out)),
                   "m3d_weight_readout")
    return out


def gaussian_lowpass(volume: torch.Tensor, sigma_zyx, is_3d: bool = True,
                     out: torch.Tensor | None = None,
                     scratch: torch.Tensor | None = None) -> torch.Tensor:
    """``cupyx.scipy.ndimage.gaussian_filter`` semantics (PixelDecoder.py:1948-1980)."""
    if volume.dim() != 3 or volume.dtype != torch.float32:
        raise ValueError("gaussian_lowpass expects a float32 [Z,Y,X] volume")
    volume = volume.contiguous()
    if out is None:
        out = torch.empty_like(volume)
    if scratch is None:
        scratch = torch.empty_like(volume)
    with torch.cuda.device(volume.device):
        _hip.check(_hip.lib().m3d_gaussian_lowpass(volume.data_ptr(), out.data_ptr(),
                                                   scratch.data_ptr(), _hip.i3(volume.shape),
                                                   _hip.f3(sigma_zyx), int(bool(is_3d)),
                                                   _stream(volume)), "m3d_gaussian_lowpass")
    return out


def decode_pixels(stack: torch.Tensor, codebook: torch.Tensor, pixel_threshold: float,
                  magnitude_threshold, background: torch.Tensor | None = None,
                  normalization: torch.Tensor | None = None, excluded_indices=(),
                  want_f32: bool = False):
    """Fused ``_decode_pixels`` (PixelDecoder.py:2523-2643).

    ``stack`` float32 ``[B,Z,Y,X]``; ``codebook`` ``[C,B]`` unit rows.  Returns
    ``decoded int16[Z,Y,X], magnitude f16, distance f16, scaled f16[B,Z,Y,X]``
    (+ float32 magnitude / distance when ``want_f32``).
    """
    if stack.dim() != 4 or stack.dtype != torch.float32:
        raise ValueError("decode_pixels expects a float32 [B,Z,Y,X] stack")
    stack = stack.contiguous()
    dev = stack.device
    B = int(stack.shape[0])
    vol_shape = tuple(stack.shape[1:])
    N = int(np.prod(vol_shape))
    cb = codebook.to(device=dev, dtype=torch.float32).contiguous()
    if cb.dim() != 2 or cb.shape[1] != B:
        raise ValueError("codebook must be [C, B]")
    bk = nm = None
    if background is not None and normalization is not None:
        bk = background.to(device=dev, dtype=torch.float32)[:B].contiguous()
        nm = normalization.to(device=dev, dtype=torch.float32)[:B].contiguous()
    ex = None
    n_ex = len(excluded_indices)
    if n_ex:
        ex = torch.as_tensor(list(excluded_indices), dtype=torch.int32, device=dev)
    decoded = torch.empty(vol_shape, dtype=torch.int16, device=dev)
    magnitude = torch.empty(vol_shape, dtype=torch.float16, device=dev)
    distance = torch.empty(vol_shape, dtype=torch.float16, device=dev)
    scaled = torch.empty((B, *vol_shape), dtype=torch.float16, device=dev)
    mag32 = dist32 = None
    if want_f32:
        mag32 = torch.empty(vol_shape, dtype=torch.float32, device=dev)
        dist32 = torch.empty(vol_shape, dtype=torch.float32, device=dev)
    thr32 = float(np.float32(pixel_threshold))
    with torch.cuda.device(dev):
        _hip.check(
            _hip.lib().m3d_decode_pixels(
                stack.data_ptr(), B, N, _hip.ptr(bk), _hip.ptr(nm), cb.data_ptr(), int(cb.shape[0]),
                thr32, float(magnitude_threshold[0]), float(magnitude_threshold[1]),
                _hip.ptr(ex), n_ex, decoded.data_ptr(), magnitude.data_ptr(), distance.data_ptr(),
                scaled.data_ptr(), _hip.ptr(mag32), _hip.ptr(dist32), _stream(stack),
            ),
            "m3d_decode_pixels",
        )
    if want_f32:
        return decoded, magnitude, distance, scaled, mag32, dist32
    return decoded, magnitude, distance, scaled


def rlgc_debug_split(image: torch.Tensor, seed: int, iteration: int) -> torch.Tensor:
    """The Philox split of one RLGC iteration (parity hook for tests)."""
    image = image.contiguous()
    code = _hip.M3D_U16 if image.dtype == torch.uint16 else _hip.M3D_F32
    if code == _hip.M3D_F32 and image.dtype != torch.float32:
        raise ValueError("image must be uint16 or float32")
    out = torch.empty(image.shape, dtype=torch.float32, device=image.device)
    with torch.cuda.device(image.device):
        _hip.check(_hip.lib().m3d_rlgc_debug_split(image.data_ptr(), code, image.numel(),
                                                   int(seed) & 0xFFFFFFFFFFFFFFFF, int(iteration),
                                                   out.data_ptr(), _stream(image)),
                   "m3d_rlgc_debug_split")
    return out
