"""RLGC deconvolution on MI355X - host side of ``m3d_rlgc_*`` (csrc/m3d_rlgc.hip).

Mirrors the public surface of the reference's ``utils/rlgc.py`` (same names,
arguments, defaults, return types and error behaviour: ``chunked_rlgc`` :1021,
``rlgc`` :497, ``clear_rlgc_caches`` :39 and the geometry helpers) so callers
switch by changing the import.  Differences that are visible on purpose:

* arrays may be NumPy (reference behaviour: host in, host out) **or** torch
  tensors already in HBM (device in, device out, nothing crosses PCIe);
* the 50:50 photon split is a counter-based Philox draw instead of a cuRAND
  stream (``rlgc.py:564,619``) - statistically identical, reproducible from
  ``rng_seed`` on any GPU, and regenerated in registers instead of stored;
* padding, OTF construction, the iteration and the crop all run inside one
  C-ABI call on a cached plan; 288 GB of HBM make the lateral-tiling route a
  fallback that is only taken when ``crop_yx`` asks for it or on a real OOM.
"""

from __future__ import annotations

import gc
import logging
import secrets
from collections.abc import Callable
from dataclasses import dataclass, field

import numpy as np

from .. import _hip


# ----------------------------------------------------------------------------
# geometry helpers (host integers; rlgc.py:72-200)
# ----------------------------------------------------------------------------
def next_gpu_fft_size(x: int) -> int:
    """Smallest FFT length >= ``x`` with prime factors in {2, 3} (rlgc.py:72-97)."""
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


def _axis_linear_fft_padding(length: int, psf_support: int, *, halo_multiplier: int = 1):
    """(before, after) padding of one axis: PSF halo + FFT-friendly extra (rlgc.py:100-131)."""
    halo = max((int(psf_support) // 2) * int(halo_multiplier), 0)
    with_halo = length + 2 * halo
    extra = next_gpu_fft_size(with_halo) - with_halo
    return halo + extra // 2, halo + extra - extra // 2


def pad_widths_zyx(image_shape, psf_shape, pad_yx: bool = True):
    """Per-axis pad widths used by the device plan (rlgc.py:163-171)."""
    pad_z = _axis_linear_fft_padding(image_shape[0], psf_shape[0])
    if pad_yx:
        return (
            pad_z,
            _axis_linear_fft_padding(image_shape[1], psf_shape[1]),
            _axis_linear_fft_padding(image_shape[2], psf_shape[2]),
        )
    return (pad_z, (0, 0), (0, 0))


def pad_for_linear_fft(image: np.ndarray, psf_shape, pad_yx: bool = True):
    """Host-side symmetric pad kept for API parity (rlgc.py:134-173).

    The device solver never materialises this array (the mask zeroes the pad,
    rlgc.py:588); it exists so code written against the reference keeps working.
    """
    if image.ndim != 3:
        raise ValueError(f"Expected 3D input, got shape {image.shape!r}")
    pad_width = pad_widths_zyx(image.shape, psf_shape, pad_yx)
    return np.pad(image, pad_width, mode="symmetric"), pad_width


def remove_padding_zyx(padded, pad_width):
    """Crop the padding added by :func:`pad_for_linear_fft` (rlgc.py:176-200)."""
    sl = []
    for axis, (before, after) in enumerate(pad_width):
        stop = padded.shape[axis] - after if after > 0 else None
        sl.append(slice(before, stop))
    return padded[tuple(sl)]


# ----------------------------------------------------------------------------
# plan cache (replaces _fft_cache_3d + the cuFFT plan cache, rlgc.py:36-69)
# ----------------------------------------------------------------------------
@dataclass
class _Plan:
    handle: int
    workspace: object  # torch.uint8 tensor
    workspace_bytes: int


_plan_cache: dict[tuple, _Plan] = {}


def _get_plan(device_index: int, image_shape, psf_shape, pad_yx: bool) -> _Plan:
    import ctypes

    import torch

    # one plan (= one workspace) per stream: solves issued on different streams may overlap
    with torch.cuda.device(device_index):
        stream = int(torch.cuda.current_stream().cuda_stream)
    key = (int(device_index), tuple(image_shape), tuple(psf_shape), bool(pad_yx), stream)
    plan = _plan_cache.get(key)
    if plan is None:
        lib = _hip.lib()
        handle = ctypes.c_void_p()
        with torch.cuda.device(device_index):
            _hip.check(
                lib.m3d_rlgc_plan_create(
                    _hip.i3(image_shape), _hip.i3(psf_shape), int(pad_yx), ctypes.byref(handle)
                ),
                "m3d_rlgc_plan_create",
            )
            nbytes = int(lib.m3d_rlgc_plan_workspace_bytes(handle))
            try:
                workspace = torch.empty(nbytes, dtype=torch.uint8, device=f"cuda:{device_index}")
            except Exception:
                lib.m3d_rlgc_plan_destroy(handle)
                raise
        plan = _Plan(handle=handle.value, workspace=workspace, workspace_bytes=nbytes)
        _plan_cache[key] = plan
    return plan


def set_fft_engine(engine: str = "fused") -> None:
    """Choose the FFT engine for plans created from now on: ``"fused"`` (in-LDS convolution
    kernels, the default) or ``"rocfft"`` (library fallback, kept for cross-checks)."""
    if engine not in ("fused", "rocfft"):
        raise ValueError("engine must be 'fused' or 'rocfft'")
    clear_rlgc_caches()
    _hip.check(_hip.lib().m3d_set_fft_engine(0 if engine == "fused" else 1), "m3d_set_fft_engine")


def clear_rlgc_caches(clear_memory_pool: bool = False) -> None:
    """Drop cached rocFFT plans + workspaces (rlgc.py:39-69)."""
    lib = _hip.lib() if _plan_cache else None
    for plan in _plan_cache.values():
        lib.m3d_rlgc_plan_destroy(plan.handle)
    _plan_cache.clear()
    for engine in _fft_cache_3d.values():
        engine.close()
    _fft_cache_3d.clear()
    if clear_memory_pool:
        import torch

        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            torch.cuda.empty_cache()


# ----------------------------------------------------------------------------
# the solver's building blocks as stand-alone device functions (rlgc.py:23-31, 203-412)
#
# ``rlgc`` / ``chunked_rlgc`` run fused versions of this arithmetic inside ``m3d_rlgc_run``; these keep the
# reference's module-level names callable on their own.  Arrays: float32 torch tensors in HBM (NumPy arrays
# are copied to the current device and the result comes back as NumPy, like the host-in / host-out solver).
# ----------------------------------------------------------------------------
def _dev_in(array, dtype=None):
    """(tensor in HBM, came_from_numpy)"""
    import torch

    if isinstance(array, torch.Tensor):
        if not array.is_cuda:
            raise ValueError("expected a tensor in GPU memory (or a NumPy array)")
        t = array if dtype is None or array.dtype == dtype else array.to(dtype)
        return t.contiguous(), False
    device = torch.device("cuda", torch.cuda.current_device())
    return _as_device(np.asarray(array), device, dtype), True


def _dev_out(tensor, as_numpy: bool):
    return tensor.cpu().numpy() if as_numpy else tensor


def _pads3(pad_width):
    pads = [(int(a), int(b)) for a, b in pad_width]
    if len(pads) != 3:
        raise ValueError("pad_width must hold three (before, after) pairs")
    return _hip.i3([p[0] for p in pads]), _hip.i3([p[1] for p in pads])


def filter_update(recon, HTratio, consensus_map, out=None):
    """Consensus-gated multiplicative update (``ElementwiseKernel`` of rlgc.py:23-31):
    ``out = recon if consensus_map < 0 else recon * HTratio``."""
    import torch

    r, host = _dev_in(recon, torch.float32)
    h, _ = _dev_in(HTratio, torch.float32)
    c, _ = _dev_in(consensus_map, torch.float32)
    if r.shape != h.shape or r.shape != c.shape:
        raise ValueError("recon, HTratio and consensus_map must have the same shape")
    res = torch.empty_like(r) if out is None else out
    _hip.check(_hip.lib().m3d_filter_update(r.data_ptr(), h.data_ptr(), c.data_ptr(), res.data_ptr(), r.numel(),
                                            _hip.current_stream(r.device)), "m3d_filter_update")
    return _dev_out(res, host and out is None)


def _symmetric_padded_axis_indices(length: int, pad_before: int, pad_after: int):
    """Source indices of the left and right pad regions of one symmetrically extended axis (rlgc.py:203-227)."""
    observed = np.arange(pad_before, length - pad_after, dtype=np.int64)
    padded = np.pad(observed, (pad_before, pad_after), mode="symmetric")
    return padded[:pad_before], padded[length - pad_after:]


def enforce_symmetric_boundary(image, pad_width) -> None:
    """Overwrite the pad regions with the symmetric extension of the observed region, in place
    (rlgc.py:230-271; the three axis passes compose to one gather from the core)."""
    import torch

    if not (isinstance(image, torch.Tensor) and image.is_cuda and image.dtype == torch.float32
            and image.is_contiguous() and image.dim() == 3):
        raise ValueError("enforce_symmetric_boundary works in place on a contiguous float32 (Z, Y, X) tensor in HBM")
    before, after = _pads3(pad_width)
    _hip.check(_hip.lib().m3d_enforce_symmetric_boundary(image.data_ptr(), _hip.i3(image.shape), before, after,
                                                         _hip.current_stream(image.device)),
               "m3d_enforce_symmetric_boundary")


def pad_psf(psf_temp, image_shape, normalize: bool = True):
    """Zero-embed and centre a PSF in ``image_shape`` (the six rolls + ifftshift of rlgc.py:274-313), unit sum."""
    import torch

    psf, host = _dev_in(psf_temp, torch.float32)
    if psf.dim() == 2:
        psf = psf.unsqueeze(0).contiguous()
    shape = tuple(int(v) for v in image_shape)
    lib = _hip.lib()
    out = torch.empty(shape, dtype=torch.float32, device=psf.device)
    nbytes = int(lib.m3d_pad_psf_workspace_bytes())
    work = torch.empty(nbytes, dtype=torch.uint8, device=psf.device)
    _hip.check(lib.m3d_pad_psf(psf.data_ptr(), _hip.i3(psf.shape), out.data_ptr(), _hip.i3(shape), int(bool(normalize)),
                               work.data_ptr(), nbytes, _hip.current_stream(psf.device)), "m3d_pad_psf")
    return _dev_out(out, host)


class TransferFunction:
    """``rfftn`` of a padded PSF as the FFT engine keeps it (tiled, digit-reversed half spectrum in HBM).
    Stands where the reference passes a complex64 array ``H`` to ``fft_conv``; ``conj()`` and ``abs2()`` give
    the views rlgc.py:583-585 calls ``otfT`` and ``otfotfT`` (no data is copied: the Z pass conjugates /
    squares on the fly)."""

    def __init__(self, engine, spectrum, mode: int = 0):
        self._engine, self._spectrum, self.mode = engine, spectrum, int(mode)

    @property
    def shape(self):
        return self._engine.shape

    def conj(self):
        if self.mode == 2:
            return self
        return TransferFunction(self._engine, self._spectrum, 1 - self.mode)

    def abs2(self):
        return TransferFunction(self._engine, self._spectrum, 2)


class _FFTConvEngine:
    def __init__(self, shape, device):
        import ctypes

        import torch

        self.shape = tuple(int(v) for v in shape)
        self.device = device
        self._lib = _hip.lib()
        handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _hip.check(self._lib.m3d_fftconv_create(_hip.i3(self.shape), ctypes.byref(handle)), "m3d_fftconv_create")
            self.handle = handle
            self.spectrum_bytes = int(self._lib.m3d_fftconv_spectrum_bytes(handle))
            self.scratch = torch.empty(self.spectrum_bytes, dtype=torch.uint8, device=device)

    def close(self):
        if self.handle:
            self._lib.m3d_fftconv_destroy(self.handle)
            self.handle = None


# (device index, z, y, x) -> engine with its scratch spectrum: the reference's cached c64 buffer (rlgc.py:36, 341-346)
_fft_cache_3d: dict[tuple[int, int, int, int], _FFTConvEngine] = {}


def _engine_for(shape, device) -> _FFTConvEngine:
    key = (int(device.index if device.index is not None else 0), *[int(v) for v in shape])
    eng = _fft_cache_3d.get(key)
    if eng is None:
        eng = _fft_cache_3d[key] = _FFTConvEngine(shape, device)
    return eng


def transfer_function(psf_padded) -> TransferFunction:
    """``cp.fft.rfftn(psf)`` of rlgc.py:582 for a PSF already padded and centred (``pad_psf``)."""
    import torch

    psf, _ = _dev_in(psf_padded, torch.float32)
    eng = _engine_for(psf.shape, psf.device)
    spectrum = torch.empty(eng.spectrum_bytes, dtype=torch.uint8, device=psf.device)
    _hip.check(eng._lib.m3d_fftconv_otf(eng.handle, psf.data_ptr(), spectrum.data_ptr(), _hip.current_stream(psf.device)),
               "m3d_fftconv_otf")
    return TransferFunction(eng, spectrum)


def fft_conv(image, H: TransferFunction, shape=None):
    """``irfftn(rfftn(image) * H, s=shape)`` (rlgc.py:316-350) in five fused passes; ``H`` comes from
    ``transfer_function`` (optionally ``.conj()`` / ``.abs2()``).  No clipping, float32 result."""
    import torch

    if not isinstance(H, TransferFunction):
        raise TypeError("H must be a TransferFunction (see transfer_function): the engine keeps spectra in its own layout")
    img, host = _dev_in(image, torch.float32)
    if shape is not None and tuple(int(v) for v in shape) != tuple(img.shape):
        raise ValueError("shape must equal image.shape")
    if tuple(img.shape) != H.shape:
        raise ValueError(f"image shape {tuple(img.shape)} does not match the transfer function's {H.shape}")
    eng = H._engine
    out = torch.empty_like(img)
    _hip.check(eng._lib.m3d_fftconv_apply(eng.handle, img.data_ptr(), H._spectrum.data_ptr(), H.mode,
                                          eng.scratch.data_ptr(), out.data_ptr(), _hip.current_stream(img.device)),
               "m3d_fftconv_apply")
    return _dev_out(out, host)


def _observed_region_mask(shape, pad_width):
    """float32 mask: 1 on the observed image, 0 in the padding (rlgc.py:353-379); a tensor on the current device."""
    import torch

    shape = tuple(int(v) for v in shape)
    before, after = _pads3(pad_width)
    mask = torch.empty(shape, dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
    _hip.check(_hip.lib().m3d_observed_region_mask(mask.data_ptr(), _hip.i3(shape), before, after,
                                                   _hip.current_stream(mask.device)), "m3d_observed_region_mask")
    return mask


def kl_div(p, q, mask=None) -> float:
    """Normalised, optionally masked Kullback-Leibler divergence with eps = 1e-4 and NaN terms dropped
    (rlgc.py:382-412).  One device pass (three double-precision sums), one host read-back - the reference's
    Python float."""
    import ctypes

    import torch

    pt, _ = _dev_in(p, torch.float32)
    qt, _ = _dev_in(q, torch.float32)
    if pt.shape != qt.shape:
        raise ValueError("p and q must have the same shape")
    mt = None
    if mask is not None:
        mt, _ = _dev_in(mask, torch.float32)
        if mt.shape != pt.shape:
            raise ValueError("mask must have the shape of p")
    lib = _hip.lib()
    nbytes = int(lib.m3d_kl_div_workspace_bytes())
    work = torch.empty(nbytes, dtype=torch.uint8, device=pt.device)
    result = ctypes.c_double(0.0)
    _hip.check(lib.m3d_kl_div(pt.data_ptr(), qt.data_ptr(), None if mt is None else mt.data_ptr(), pt.numel(),
                              work.data_ptr(), nbytes, ctypes.byref(result), _hip.current_stream(pt.device)),
               "m3d_kl_div")
    return float(result.value)


# ----------------------------------------------------------------------------
# solver
# ----------------------------------------------------------------------------
@dataclass
class RLGCInfo:
    """Diagnostics of one solver call (the reference only logs these, rlgc.py:699-712)."""

    iterations: int = 0
    stop_reason: str = ""
    padded_shape: tuple = ()
    kld_image: list = field(default_factory=list)
    kld_split1: list = field(default_factory=list)
    kld_split2: list = field(default_factory=list)
    updated_fraction: list = field(default_factory=list)
    max_relative_delta: list = field(default_factory=list)
    update_min: list = field(default_factory=list)
    update_max: list = field(default_factory=list)
    conv_ms_total: float = 0.0   # only with profile=True: device time inside fft_conv
    conv_calls: int = 0
    fused_engine: bool = True
    engine_flags: int = 0        # _hip.ENGINE_* bits: which kernel families ran (diagnostic, asserted by tests)
    kernel_ms: tuple = ()        # profile + fused engine: x fwd, y fwd, z conv, y inv, x inv
    loop_ms: float = 0.0         # profile: device time of the whole iteration loop


def _as_device(array, device, dtype=None):
    import torch

    if isinstance(array, torch.Tensor):
        t = array.to(device=device)
    else:
        arr = np.ascontiguousarray(array)
        if arr.dtype == np.uint16:  # kept as 2-byte counts; the kernels convert on load
            return torch.from_numpy(arr).to(device).contiguous()
        t = torch.from_numpy(arr).to(device)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def rlgc_device(
    image,
    psf,
    *,
    safe_mode: bool = True,
    limit: float = 0.01,
    max_delta: float = 0.001,
    pad_yx: bool = True,
    rng_seed: int | None = 42,
    normalize_psf: bool = True,
    max_iterations: int = 0,
    injected_splits=None,
    image_is_u16: bool | None = None,
    out=None,
    profile: bool | str = False,
):
    """Device-resident RLGC: torch tensors in HBM in, float32 tensor in HBM out.

    ``image`` is ``[Z,Y,X]`` float32, or int16/uint16 storage holding uint16
    counts (``image_is_u16``); ``psf`` float32 ``[Zp,Yp,Xp]``.  Returns
    ``(deconvolved float32[Z,Y,X], RLGCInfo)``.  ``profile``: ``True`` records HIP events around every FFT pass
    (``conv_ms_total``, ``kernel_ms``, ``loop_ms``), ``"loop"`` only the two around the iteration loop (``loop_ms``).
    """
    import ctypes

    import torch

    if image.dim() != 3 or psf.dim() != 3:
        raise ValueError(f"Expected 3D image and PSF, got {tuple(image.shape)} / {tuple(psf.shape)}")
    if not image.is_cuda:
        raise ValueError("rlgc_device expects the image in GPU memory")
    dev = image.device
    if image_is_u16 is None:
        image_is_u16 = image.dtype == torch.uint16
    if image_is_u16:
        if image.element_size() != 2:
            raise ValueError("uint16 image must use 2-byte storage")
        dtype_code = _hip.M3D_U16
    else:
        if image.dtype != torch.float32:
            image = image.to(torch.float32)
        dtype_code = _hip.M3D_F32
    image = image.contiguous()
    psf = psf.to(device=dev, dtype=torch.float32).contiguous()

    plan = _get_plan(dev.index or 0, image.shape, psf.shape, pad_yx)
    if out is None:
        out = torch.empty(tuple(image.shape), dtype=torch.float32, device=dev)
    params = _hip.RLGCParams(
        safe_mode=int(bool(safe_mode)),
        limit=float(limit),
        max_delta=float(max_delta),
        seed=int(secrets.randbits(63) if rng_seed is None else rng_seed) & 0xFFFFFFFFFFFFFFFF,
        normalize_psf=int(bool(normalize_psf)),
        max_iterations=int(max_iterations),
        profile=2 if profile == "loop" else int(bool(profile)),
    )
    n_inj = 0
    inj_ptr = None
    if injected_splits is not None:
        injected_splits = injected_splits.to(device=dev, dtype=torch.float32).contiguous()
        if tuple(injected_splits.shape[1:]) != tuple(image.shape):
            raise ValueError("injected_splits must be [n, Z, Y, X]")
        n_inj = int(injected_splits.shape[0])
        inj_ptr = injected_splits.data_ptr()
    result = _hip.RLGCResult()
    with torch.cuda.device(dev):
        status = _hip.lib().m3d_rlgc_run(
            plan.handle,
            image.data_ptr(),
            dtype_code,
            psf.data_ptr(),
            out.data_ptr(),
            plan.workspace.data_ptr(),
            plan.workspace_bytes,
            ctypes.byref(params),
            inj_ptr,
            n_inj,
            ctypes.byref(result),
            _hip.current_stream(dev),
        )
    _hip.check(status, "m3d_rlgc_run")
    n = int(result.n_trace)
    m = min(int(result.iterations), _hip.TRACE_CAP)
    pads = pad_widths_zyx(image.shape, psf.shape, pad_yx)
    info = RLGCInfo(
        iterations=int(result.iterations),
        stop_reason=_hip.STOP_REASONS.get(int(result.stop_reason), "?"),
        padded_shape=tuple(int(s + b + a) for s, (b, a) in zip(image.shape, pads)),
        kld_image=list(result.kld_image[:n]),
        kld_split1=list(result.kld_split1[:n]),
        kld_split2=list(result.kld_split2[:n]),
        updated_fraction=list(result.updated_fraction[:m]),
        max_relative_delta=list(result.max_relative_delta[:m]),
        update_min=list(result.update_min[:m]),
        update_max=list(result.update_max[:m]),
        conv_ms_total=float(result.conv_ms_total),
        conv_calls=int(result.conv_calls),
        fused_engine=bool(result.fused_engine),
        engine_flags=int(result.engine_flags),
        kernel_ms=tuple(float(v) for v in result.kernel_ms),
        loop_ms=float(result.loop_ms),
    )
    return out, info


def _log_run(logger, log_tag, info: RLGCInfo, image_shape, psf_shape, safe_mode, pad_yx):
    logger.info(
        "%ssolver_started image_shape=%s padded_shape=%s psf_shape=%s reference_core=non_accelerated safe_mode=%s pad_yx=%s",
        log_tag, tuple(image_shape), info.padded_shape, tuple(psf_shape), safe_mode, pad_yx,
    )
    for i in range(len(info.updated_fraction)):
        logger.info(
            "%siteration=%03d kld_image=%.6f kld_split1=%.6f kld_split2=%.6f update_min=%.3f update_max=%.3f updated_fraction=%.5f max_relative_delta=%.5f",
            log_tag, i + 1, info.kld_image[i], info.kld_split1[i], info.kld_split2[i],
            info.update_min[i], info.update_max[i], info.updated_fraction[i],
            info.max_relative_delta[i],
        )
    logger.info("%sstop=%s iterations=%d", log_tag, info.stop_reason, info.iterations)


def rlgc(
    image,
    psf,
    gpu_id: int = 0,
    safe_mode: bool = True,
    limit: float = 0.01,
    max_delta: float = 0.001,
    pad_yx: bool = True,
    rng_seed: int | None = 42,
    normalize_psf: bool = True,
    release_memory: bool = True,
    logger: logging.Logger | None = None,
    log_prefix: str = "",
    *,
    return_info: bool = False,
    max_iterations: int = 0,
):
    """Richardson-Lucy gradient-consensus deconvolution (rlgc.py:497-757).

    NumPy in -> NumPy float32 out like the reference; torch tensors in HBM are
    accepted and returned as tensors (no host round trip).
    """
    import torch

    is_tensor = isinstance(image, torch.Tensor)
    if psf.ndim == 2:
        psf = psf[None, ...]
    if image.ndim == 2:
        image = image[None, ...]
    if image.ndim != 3:
        raise ValueError(f"Expected 3D input, got shape {tuple(image.shape)!r}")
    device = torch.device("cuda", int(gpu_id))
    cleanup = bool(release_memory)
    try:
        image_t = _as_device(image, device)
        psf_t = _as_device(psf, device, torch.float32)
        out, info = rlgc_device(
            image_t, psf_t, safe_mode=safe_mode, limit=limit, max_delta=max_delta, pad_yx=pad_yx,
            rng_seed=rng_seed, normalize_psf=normalize_psf, max_iterations=max_iterations,
        )
        if logger is not None and logger.isEnabledFor(logging.INFO):
            _log_run(logger, f"{log_prefix} " if log_prefix else "", info, image.shape,
                     psf.shape, safe_mode, pad_yx)
        result = out if is_tensor else out.cpu().numpy().astype(np.float32, copy=False)
        return (result, info) if return_info else result
    except Exception:
        cleanup = True
        raise
    finally:
        if cleanup:
            clear_rlgc_caches(clear_memory_pool=True)


# ----------------------------------------------------------------------------
# routing: full frame vs lateral tiles, OOM fallback (rlgc.py:415-494, 760-1157)
# ----------------------------------------------------------------------------
def _child_log_prefix(base_prefix: str, suffix: str) -> str:
    return suffix if not base_prefix else f"{base_prefix} {suffix}"


def _resolve_tiled_axis_geometry(requested_crop, image_size, psf_support, axis_name):
    """Retained size and discarded halo (= full PSF support) for one axis (rlgc.py:434-467)."""
    if requested_crop <= 0:
        raise ValueError(f"{axis_name} must be greater than 0 for tiled 3D RLGC.")
    retained = min(int(requested_crop), int(image_size))
    if retained >= image_size:
        return retained, 0
    return retained, int(psf_support)


def _axis_retained_bounds(retained_size: int, image_size: int):
    """Non-overlapping bounds covering ``[0, image_size)`` (rlgc.py:470-494)."""
    if retained_size <= 0:
        raise ValueError("retained_size must be greater than 0.")
    bounds, start = [], 0
    while start < image_size:
        stop = min(start + retained_size, image_size)
        bounds.append((start, stop))
        start = stop
    return bounds


def _is_gpu_memory_error(exc: BaseException) -> bool:
    """Allocation failures that trigger the lateral-chunk retry (rlgc.py:760-780)."""
    if isinstance(exc, MemoryError):
        return True
    try:
        import torch

        if isinstance(exc, torch.cuda.OutOfMemoryError):
            return True
    except Exception:
        pass
    message = str(exc).lower()
    return "out of memory" in message or "oom" in message


def _chunked_rlgc_once(
    image,
    psf,
    gpu_id: int = 0,
    crop_yx: int = 2048,
    crop_z: int | None = None,
    safe_mode: bool = True,
    limit: float = 0.01,
    max_delta: float = 0.001,
    rng_seed: int | None = 42,
    normalize_psf: bool = True,
    verbose: int = 0,
    release_memory: bool = True,
    logger: logging.Logger | None = None,
    log_prefix: str = "",
):
    """One attempt at full-frame or laterally tiled RLGC (rlgc.py:783-1018)."""
    import torch

    if crop_yx <= 0:
        raise ValueError("crop_yx must be greater than 0.")
    if crop_z is not None and crop_z <= 0:
        raise ValueError("crop_z must be greater than 0 when provided.")
    is_tensor = isinstance(image, torch.Tensor)
    image_arr = image if is_tensor else np.asarray(image)
    original_ndim = image_arr.ndim
    if original_ndim == 2:
        work = image_arr[None, ...]
    elif original_ndim == 3:
        work = image_arr
    else:
        raise ValueError(f"Expected a 2D or 3D image, got shape {tuple(image_arr.shape)}")
    psf_arr = psf if isinstance(psf, torch.Tensor) else np.asarray(psf)
    if psf_arr.ndim not in (2, 3):
        raise ValueError(f"Expected a 2D or 3D PSF, got shape {tuple(psf_arr.shape)}")
    psf_shape = tuple(psf_arr.shape) if psf_arr.ndim == 3 else (1, *psf_arr.shape)
    eff_crop_z = work.shape[0] if crop_z is None else crop_z
    info_enabled = logger is not None and logger.isEnabledFor(logging.INFO)
    tag = f"{log_prefix} " if log_prefix else ""
    common = dict(safe_mode=safe_mode, limit=limit, max_delta=max_delta,
                  normalize_psf=normalize_psf, release_memory=False, logger=logger)

    if eff_crop_z >= work.shape[0] and crop_yx >= work.shape[-2] and crop_yx >= work.shape[-1]:
        if info_enabled:
            logger.info("%spath=3d_fullframe image_shape=%s psf_shape=%s", tag,
                        tuple(int(v) for v in image_arr.shape), tuple(int(v) for v in psf_shape))
        output = rlgc(image_arr, psf_arr, gpu_id, pad_yx=True, rng_seed=rng_seed,
                      log_prefix=_child_log_prefix(log_prefix, "path=3d_fullframe"), **common)
        if original_ndim == 2 and output.ndim == 3 and output.shape[0] == 1:
            output = output[0]
    else:
        full = tuple(work.shape)
        if eff_crop_z >= full[0]:
            retained_z, pad_z = full[0], 0
        else:
            pad_z = int(psf_shape[0])
            if eff_crop_z <= 2 * pad_z:
                raise ValueError(
                    "crop_z must be larger than twice the axial processing halo "
                    f"({2 * pad_z} for PSF support {int(psf_shape[0])})."
                )
            retained_z = int(eff_crop_z) - 2 * pad_z
        retained_y, pad_y = _resolve_tiled_axis_geometry(crop_yx, full[1], int(psf_shape[-2]), "crop_yx")
        retained_x, pad_x = _resolve_tiled_axis_geometry(crop_yx, full[2], int(psf_shape[-1]), "crop_yx")
        if is_tensor:
            output = torch.zeros(full, dtype=torch.float32, device=work.device)
        else:
            output = np.zeros(full, dtype=np.float32)
        bz = _axis_retained_bounds(retained_z, full[0])
        by = _axis_retained_bounds(retained_y, full[1])
        bx = _axis_retained_bounds(retained_x, full[2])
        tiles = [(z, y, x) for x in bx for y in by for z in bz]  # x outer, z inner (:928-933)
        if info_enabled:
            logger.info(
                "%spath=3d_tiled image_shape=%s psf_shape=%s retained_shape=%s processing_halo=%s num_tiles=%d",
                tag, full, tuple(int(v) for v in psf_shape), (retained_z, retained_y, retained_x),
                (pad_z, pad_y, pad_x), len(tiles),
            )
        iterator = enumerate(tiles)
        if verbose >= 1:
            try:
                from rich.progress import track

                iterator = track(enumerate(tiles), description="Chunks", total=len(tiles), transient=True)
            except Exception:
                pass
        for tile_idx, ((z0, z1), (y0, y1), (x0, x1)) in iterator:
            zc0, zc1 = max(z0 - pad_z, 0), min(z1 + pad_z, full[0])
            yc0, yc1 = max(y0 - pad_y, 0), min(y1 + pad_y, full[1])
            xc0, xc1 = max(x0 - pad_x, 0), min(x1 + pad_x, full[2])
            crop = work[zc0:zc1, yc0:yc1, xc0:xc1]
            dec = rlgc(
                crop.contiguous() if is_tensor else np.ascontiguousarray(crop), psf_arr, gpu_id,
                rng_seed=None if rng_seed is None else rng_seed + tile_idx,
                log_prefix=_child_log_prefix(log_prefix, f"path=3d_tiled tile={tile_idx:04d}"),
                **common,
            )
            output[z0:z1, y0:y1, x0:x1] = dec[
                z0 - zc0 : z0 - zc0 + (z1 - z0),
                y0 - yc0 : y0 - yc0 + (y1 - y0),
                x0 - xc0 : x0 - xc0 + (x1 - x0),
            ]
        gc.collect()
        if original_ndim == 2:
            output = output[0]
    if release_memory:
        clear_rlgc_caches(clear_memory_pool=True)
    return output


def chunked_rlgc(
    image,
    psf,
    gpu_id: int = 0,
    crop_yx: int = 2048,
    crop_z: int | None = None,
    safe_mode: bool = True,
    limit: float = 0.01,
    max_delta: float = 0.001,
    rng_seed: int | None = 42,
    normalize_psf: bool = True,
    verbose: int = 0,
    release_memory: bool = True,
    logger: logging.Logger | None = None,
    log_prefix: str = "",
    on_successful_crop_yx: Callable[[int], None] | None = None,
):
    """RLGC with automatic lateral-chunk fallback on GPU OOM (rlgc.py:1021-1157)."""
    if crop_z is not None:
        raise ValueError("RLGC no longer supports axial chunking; leave crop_z=None.")
    ndim = image.ndim
    if ndim == 2:
        max_yx = max(image.shape)
    elif ndim == 3:
        max_yx = max(image.shape[-2:])
    else:
        raise ValueError(f"Expected a 2D or 3D image, got shape {tuple(image.shape)}")
    fallback_step = 128
    min_crop_yx = max(int(psf.shape[-2]), int(psf.shape[-1]))
    attempted = min(int(crop_yx), int(max_yx))
    tag = f"{log_prefix} " if log_prefix else ""
    while True:
        try:
            if logger is not None and logger.isEnabledFor(logging.INFO):
                logger.info("%sattempt_rlgc_crop_yx=%d requested_crop_yx=%d", tag, attempted, crop_yx)
            output = _chunked_rlgc_once(
                image=image, psf=psf, gpu_id=gpu_id, crop_yx=attempted, crop_z=None,
                safe_mode=safe_mode, limit=limit, max_delta=max_delta, rng_seed=rng_seed,
                normalize_psf=normalize_psf, verbose=verbose, release_memory=release_memory,
                logger=logger, log_prefix=log_prefix,
            )
            if on_successful_crop_yx is not None:
                on_successful_crop_yx(attempted)
            if logger is not None and logger.isEnabledFor(logging.INFO):
                logger.info("%ssuccessful_rlgc_crop_yx=%d requested_crop_yx=%d", tag, attempted, crop_yx)
            return output
        except Exception as exc:
            if not _is_gpu_memory_error(exc):
                raise
            clear_rlgc_caches(clear_memory_pool=True)
            next_crop = attempted - fallback_step
            if next_crop < min_crop_yx:
                raise RuntimeError(
                    "RLGC failed due to GPU memory constraints even at the "
                    f"minimum lateral crop size {attempted}."
                ) from exc
            if logger is not None and logger.isEnabledFor(logging.WARNING):
                logger.warning("%sretry_after_gpu_oom previous_crop_yx=%d next_crop_yx=%d",
                               tag, attempted, next_crop)
            attempted = next_crop
