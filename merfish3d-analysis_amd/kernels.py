"""Thin torch-tensor wrappers over the C ABI (one function per ``m3d_*`` entry point).

These are the only callers of ``libm3d_hip.so`` besides ``utils/rlgc.py``; the
operator classes (``PixelDecoder``, ``DataRegistration``) and the L3 utility
functions are written on top of them.  Every function takes tensors that
already live in HBM and enqueues on the current torch stream.
"""

from __future__ import annotations

import os

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
                  want_f32: bool = False, prenormalized: bool = False):
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
                scaled.data_ptr(), _hip.ptr(mag32), _hip.ptr(dist32), int(bool(prenormalized)),
                _stream(stack),
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


def binom_debug_table_draw(q: torch.Tensor, hi: torch.Tensor, lo: torch.Tensor) -> torch.Tensor:
    """Binomial(64 q, 1/2) by table inversion of the 64-bit uniforms (hi:lo) on the device (parity hook)."""
    q, hi, lo = (t.contiguous() for t in (q, hi, lo))
    if not (q.dtype == hi.dtype == lo.dtype == torch.uint32) or not (q.shape == hi.shape == lo.shape):
        raise ValueError("q, hi, lo must be uint32 tensors of one shape")
    out = torch.empty(q.shape, dtype=torch.uint32, device=q.device)
    with torch.cuda.device(q.device):
        _hip.check(_hip.lib().m3d_binom_debug_table_draw(q.data_ptr(), hi.data_ptr(), lo.data_ptr(), q.numel(),
                                                         out.data_ptr(), _stream(q)), "m3d_binom_debug_table_draw")
    return out


def fft_debug_pk(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    """The packed complex primitives and small butterflies of the FFT passes on complex64 triples (parity hook):
    returns (n, 112) complex64, layout as in ``k_debug_pk`` (csrc/m3d_fftconv.hip)."""
    a, b, c = (t.contiguous() for t in (a, b, c))
    if not (a.dtype == b.dtype == c.dtype == torch.complex64) or not (a.shape == b.shape == c.shape) or a.dim() != 1:
        raise ValueError("a, b, c must be one-dimensional complex64 tensors of one length")
    out = torch.empty((a.numel(), 112), dtype=torch.complex64, device=a.device)  # kDebugPkSlots
    with torch.cuda.device(a.device):
        _hip.check(_hip.lib().m3d_fft_debug_pk(a.data_ptr(), b.data_ptr(), c.data_ptr(), a.numel(), out.data_ptr(),
                                               _stream(a)), "m3d_fft_debug_pk")
    return out


SUM_COLS = 11  # area, sz, sy, sx, szz, syy, sxx, szy, szx, syx, magnitude sum (+ B intensity sums)


def connected_component_table(decoded: torch.Tensor, intensity: torch.Tensor, magnitude: torch.Tensor,
                              distance: torch.Tensor, is_3d: bool = True):
    """CCL + per-component sums (PixelDecoder.py:2946-3042) -> small host arrays.

    ``decoded`` int16 ``[Z,Y,X]``; ``intensity`` float16/float32 ``[B,Z,Y,X]``; ``magnitude``
    and ``distance`` float16 ``[Z,Y,X]``.  Returns a dict of NumPy arrays with one row per
    connected component, ordered like scikit-image labels (raster order of first voxel):
    ``root, code, min_distance, sums[n, 11 + B]``.
    """
    dev = decoded.device
    if decoded.dtype != torch.int16 or decoded.dim() != 3:
        raise ValueError("decoded must be int16 [Z,Y,X]")
    decoded = decoded.contiguous()
    intensity = intensity.contiguous()
    magnitude = magnitude.contiguous()
    distance = distance.contiguous()
    if intensity.dtype not in (torch.float16, torch.float32):
        raise ValueError("intensity must be float16 or float32")
    if magnitude.dtype != torch.float16 or distance.dtype != torch.float16:
        raise ValueError("magnitude and distance must be float16")
    B = int(intensity.shape[0])
    shape = tuple(int(v) for v in decoded.shape)
    labels = torch.empty(shape, dtype=torch.int32, device=dev)
    slot = torch.empty(shape, dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    lib = _hip.lib()
    with torch.cuda.device(dev):
        _hip.check(lib.m3d_ccl_label(decoded.data_ptr(), _hip.i3(shape), 3 if is_3d else 2,
                                     0 if is_3d else 1, labels.data_ptr(), slot.data_ptr(),
                                     count.data_ptr(), _stream(decoded)), "m3d_ccl_label")
        n = int(count.item())
        root = torch.empty(n, dtype=torch.int32, device=dev)
        code = torch.empty(n, dtype=torch.int32, device=dev)
        mind = torch.empty(n, dtype=torch.float32, device=dev)
        sums = torch.empty((n, SUM_COLS + B), dtype=torch.float64, device=dev)
        code_dtype = _hip.M3D_F16 if intensity.dtype == torch.float16 else _hip.M3D_F32
        _hip.check(lib.m3d_region_stats(decoded.data_ptr(), labels.data_ptr(), slot.data_ptr(),
                                        _hip.i3(shape), B, intensity.data_ptr(), code_dtype,
                                        magnitude.data_ptr(), distance.data_ptr(), n,
                                        _hip.ptr(root) if n else None, _hip.ptr(code) if n else None,
                                        _hip.ptr(mind) if n else None, _hip.ptr(sums) if n else None,
                                        _stream(decoded)), "m3d_region_stats")
    root_h = root.cpu().numpy()
    order = np.argsort(root_h, kind="stable")
    return {
        "root": root_h[order],
        "code": code.cpu().numpy()[order],
        "min_distance": mind.cpu().numpy()[order],
        "sums": sums.cpu().numpy()[order],
        "labels": labels,
        "slot": slot,
    }


_MODES = {"constant": 0, "nearest": 1}


def _image_dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.uint16:
        return _hip.M3D_U16
    if t.dtype == torch.float32:
        return _hip.M3D_F32
    raise ValueError(f"volume must be uint16 or float32, got {t.dtype}")


def max_z_projection(volume: torch.Tensor) -> torch.Tensor:
    """``cp.max(image, axis=0)`` as float32 (multiview_registration.py:57-77)."""
    volume = volume.contiguous()
    out = torch.empty(tuple(volume.shape[1:]), dtype=torch.float32, device=volume.device)
    with torch.cuda.device(volume.device):
        _hip.check(_hip.lib().m3d_max_z(volume.data_ptr(), _image_dtype_code(volume),
                                        _hip.i3(volume.shape), out.data_ptr(), _stream(volume)),
                   "m3d_max_z")
    return out


def affine_warp(volume: torch.Tensor, matrix_px, offset_px, output_shape, mode: str = "constant",
                cval: float = 0.0) -> torch.Tensor:
    """``ndimage.affine_transform(order=1)`` with output dtype = input dtype (:883-891)."""
    import ctypes

    if mode not in _MODES:
        raise NotImplementedError(f"boundary mode {mode!r} is not supported (constant, nearest)")
    volume = volume.contiguous()
    out = torch.empty(tuple(int(v) for v in output_shape), dtype=volume.dtype, device=volume.device)
    m = (ctypes.c_double * 9)(*[float(v) for v in np.asarray(matrix_px, dtype=np.float64).ravel()])
    o = (ctypes.c_double * 3)(*[float(v) for v in np.asarray(offset_px, dtype=np.float64).ravel()])
    with torch.cuda.device(volume.device):
        _hip.check(_hip.lib().m3d_affine_warp(volume.data_ptr(), _image_dtype_code(volume),
                                              _hip.i3(volume.shape), m, o, out.data_ptr(),
                                              _hip.i3(out.shape), _MODES[mode], float(cval),
                                              _stream(volume)), "m3d_affine_warp")
    return out


def flow_warp(volume: torch.Tensor, flow_xyz: torch.Tensor, stride_zyx, box_start_zyx, spacing_zyx,
              origin_zyx, transform_zyx_um, output_shape, mode: str = "constant",
              cval: float = 0.0) -> torch.Tensor:
    """Affine + SOFIMA-flow resample in one pass (multiview_registration.py:1027-1101)."""
    if mode not in _MODES:
        raise NotImplementedError(f"boundary mode {mode!r} is not supported (constant, nearest)")
    volume = volume.contiguous()
    flow_xyz = flow_xyz.to(device=volume.device, dtype=torch.float32).contiguous()
    if flow_xyz.dim() != 4 or flow_xyz.shape[0] != 3:
        raise ValueError("SOFIMA flow field must have three XYZ channels.")
    out = torch.empty(tuple(int(v) for v in output_shape), dtype=volume.dtype, device=volume.device)
    T = (_hip.c_float * 16)(*[float(v) for v in np.asarray(transform_zyx_um, dtype=np.float32).ravel()])
    with torch.cuda.device(volume.device):
        _hip.check(_hip.lib().m3d_flow_warp(
            volume.data_ptr(), _image_dtype_code(volume), _hip.i3(volume.shape), out.data_ptr(),
            _hip.i3(out.shape), flow_xyz.data_ptr(), _hip.i3(flow_xyz.shape[1:]),
            _hip.f3(np.asarray(stride_zyx, dtype=np.float32)),
            _hip.f3(np.asarray(box_start_zyx, dtype=np.float32)),
            _hip.f3(np.asarray(spacing_zyx, dtype=np.float32)),
            _hip.f3(np.asarray(origin_zyx, dtype=np.float32)), T, _MODES[mode], float(cval),
            _stream(volume)), "m3d_flow_warp")
    return out


# ----------------------------------------------------------------------------
# exact order statistics (global normalisation, PixelDecoder.py:1113-1177)
# ----------------------------------------------------------------------------
def _key_to_float(key: int) -> float:
    bits = (~key) & 0xFFFFFFFF if not (key & 0x80000000) else key & 0x7FFFFFFF
    return float(np.array([bits], dtype=np.uint32).view(np.float32)[0])


class PooledValues:
    """A multiset of float32 voxels spread over several tensors (and, with ``group``, over
    several ranks), optionally transformed ``max(x - sub, 0)`` and filtered ``< / > cut``
    per tensor - the "pooled tails" of PixelDecoder.py:1113-1127 and :1144-1177."""

    def __init__(self, group=None):
        self.parts = []  # (tensor, transform, sub, filter, thr)
        self.group = group

    def add(self, values: torch.Tensor, *, sub: float | None = None, below: float | None = None,
            above: float | None = None):
        if values.dtype != torch.float32:
            raise ValueError("order statistics run on float32 volumes")
        flt, thr = (1, below) if below is not None else ((2, above) if above is not None else (0, 0.0))
        self.parts.append((values.contiguous().view(-1), 0 if sub is None else 1,
                           0.0 if sub is None else float(sub), flt, float(thr)))

    def _local_hist(self, pass_idx: int, prefix: int, device) -> torch.Tensor:
        """This rank's 256-bin histogram of radix digit ``pass_idx`` among keys matching ``prefix``."""
        hist = torch.zeros(256, dtype=torch.int64, device=device)
        lib = _hip.lib()
        for values, transform, sub, flt, thr in self.parts:
            with torch.cuda.device(values.device):
                _hip.check(lib.m3d_radix_hist(values.data_ptr(), values.numel(), transform, sub, flt,
                                              thr, pass_idx, prefix, hist.data_ptr(),
                                              _stream(values)), "m3d_radix_hist")
        return hist

    def _hist(self, pass_idx: int, prefix: int, device) -> np.ndarray:
        hist = self._local_hist(pass_idx, prefix, device)
        if self.group is not None or _dist_ready():
            import torch.distributed as dist

            if self.group is not False:
                dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=self.group or None)
        return hist.cpu().numpy()

    def count(self) -> int:
        dev = self.parts[0][0].device if self.parts else "cuda"
        return int(self._hist(0, 0, dev).sum())

    def kth(self, ks) -> list[float]:
        """Values at 0-based ranks ``ks`` of the pooled, filtered multiset (exact)."""
        dev = self.parts[0][0].device
        out = []
        for k in ks:
            prefix, rank = 0, int(k)
            for p in range(4):
                h = self._hist(p, prefix, dev)
                c = np.cumsum(h)
                b = int(np.searchsorted(c, rank, side="right"))
                if b > 255:
                    raise IndexError("rank beyond the number of pooled values")
                rank -= int(c[b - 1]) if b else 0
                prefix |= b << (24 - 8 * p)
            out.append(_key_to_float(prefix))
        return out

    def percentile(self, q: float) -> float:
        """``cp.percentile(x, q)`` (linear method, float64 result)."""
        n = self.count()
        if n == 0:
            return float("nan")
        idx = (float(q) / 100.0) * (n - 1)
        lo = int(np.floor(idx))
        hi = min(lo + 1, n - 1)
        a, b = self.kth([lo, hi]) if hi != lo else (self.kth([lo]) * 2)
        t = idx - lo
        diff = b - a
        return float(a + diff * t) if t < 0.5 else float(b - diff * (1.0 - t))

    def median(self) -> float:
        """``cp.median`` of float32 values: mean of the two middle order statistics in float32."""
        n = self.count()
        if n == 0:
            return float("nan")
        if n % 2:
            return self.kth([n // 2])[0]
        a, b = self.kth([n // 2 - 1, n // 2])
        return float((np.float32(a) + np.float32(b)) / np.float32(2.0))


def _dist_ready() -> bool:
    """True when the job runs one process per GPU under ``torch.distributed``.  A process group of ONE rank
    counts only with ``M3D_DIST_SINGLE_RANK=1``: that is how a one-GPU box drives the RCCL code path
    (device histogram -> ``all_reduce``) that the multi-GPU run takes."""
    try:
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size() > 1 or os.environ.get("M3D_DIST_SINGLE_RANK") == "1"
    except Exception:
        return False


def replace_above(volume: torch.Tensor, threshold: float, replacement: float) -> torch.Tensor:
    """In place ``v[v > threshold] = replacement`` (PixelDecoder.py:1072-1074)."""
    with torch.cuda.device(volume.device):
        _hip.check(_hip.lib().m3d_replace_above(volume.data_ptr(), volume.numel(), float(threshold),
                                                float(replacement), _stream(volume)),
                   "m3d_replace_above")
    return volume


# ----------------------------------------------------------------------------
# phase correlation building blocks (multiview_registration.py:265-301)
# ----------------------------------------------------------------------------
class FFTPlan:
    """In-place complex64 FFT of a 1..3-D tensor through ``m3d_fft_*`` (rocFFT)."""

    def __init__(self, shape, device, batch: int = 1):
        import ctypes

        self.batch = int(batch)
        self.transform_shape = tuple(int(v) for v in shape)
        self.shape = self.transform_shape if self.batch == 1 else (self.batch,) + self.transform_shape
        self.device = device
        handle = ctypes.c_void_p()
        arr = (ctypes.c_int32 * len(self.transform_shape))(*self.transform_shape)
        with torch.cuda.device(device):
            _hip.check(_hip.lib().m3d_fft_plan_create_batched(len(self.transform_shape), arr, self.batch,
                                                              ctypes.byref(handle)),
                       "m3d_fft_plan_create_batched")
        self.handle = handle.value
        nbytes = int(_hip.lib().m3d_fft_plan_workspace_bytes(self.handle))
        self.work = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        self.work_bytes = nbytes

    def execute(self, data: torch.Tensor, inverse: bool = False) -> torch.Tensor:
        shapes = (self.shape, (self.batch,) + self.transform_shape)
        if data.dtype != torch.complex64 or tuple(data.shape) not in shapes or not data.is_contiguous():
            raise ValueError("FFTPlan.execute expects a contiguous complex64 tensor of the planned shape")
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().m3d_fft_exec(self.handle, data.data_ptr(), int(bool(inverse)),
                                               self.work.data_ptr(), self.work_bytes, _stream(data)),
                       "m3d_fft_exec")
        return data

    def close(self):
        if self.handle:
            _hip.lib().m3d_fft_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def real_to_complex(volume: torch.Tensor) -> torch.Tensor:
    volume = volume.contiguous()
    out = torch.empty(volume.shape, dtype=torch.complex64, device=volume.device)
    with torch.cuda.device(volume.device):
        _hip.check(_hip.lib().m3d_real_to_complex(volume.data_ptr(), _image_dtype_code(volume),
                                                  volume.numel(), out.data_ptr(), _stream(volume)),
                   "m3d_real_to_complex")
    return out


def cross_power(src_freq: torch.Tensor, target_freq: torch.Tensor, normalize: bool = True,
                out: torch.Tensor | None = None) -> torch.Tensor:
    if out is None:
        out = torch.empty_like(src_freq)
    with torch.cuda.device(src_freq.device):
        _hip.check(_hip.lib().m3d_cross_power(src_freq.data_ptr(), target_freq.data_ptr(),
                                              src_freq.numel(), int(bool(normalize)), out.data_ptr(),
                                              _stream(src_freq)), "m3d_cross_power")
    return out


def argmax_abs(data: torch.Tensor) -> int:
    """``np.argmax(np.abs(x))`` of a complex64 tensor (flat index, first maximum)."""
    dev = data.device
    scratch = torch.empty(64 * 1024, dtype=torch.uint8, device=dev)
    idx = torch.zeros(1, dtype=torch.int64, device=dev)
    val = torch.zeros(1, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _hip.check(_hip.lib().m3d_argmax_abs(data.data_ptr(), data.numel(), scratch.data_ptr(),
                                             idx.data_ptr(), val.data_ptr(), _stream(data)),
                   "m3d_argmax_abs")
    return int(idx.item())


def contract_last(data: torch.Tensor, kernel: torch.Tensor, conj_data: bool = False) -> torch.Tensor:
    """``np.tensordot(kernel, data, axes=(1, -1))`` for complex64 ``data[..., X]``, ``kernel[K, X]``."""
    data = data.contiguous()
    kernel = kernel.to(device=data.device, dtype=torch.complex64).contiguous()
    X = int(data.shape[-1])
    rows = data.numel() // X
    Ka = int(kernel.shape[0])
    out = torch.empty((Ka, *data.shape[:-1]), dtype=torch.complex64, device=data.device)
    with torch.cuda.device(data.device):
        _hip.check(_hip.lib().m3d_contract_last(data.data_ptr(), rows, X, kernel.data_ptr(), Ka,
                                                int(bool(conj_data)), out.data_ptr(), _stream(data)),
                   "m3d_contract_last")
    return out


def box_corr_sums(a: torch.Tensor, b: torch.Tensor, lo, hi) -> np.ndarray:
    """(n, sum a, sum b, sum aa, sum bb, sum ab) over the box ``[lo, hi)`` of two f32 volumes."""
    sums = torch.zeros(6, dtype=torch.float64, device=a.device)
    with torch.cuda.device(a.device):
        _hip.check(_hip.lib().m3d_box_corr_sums(a.data_ptr(), b.data_ptr(), _hip.i3(a.shape),
                                                _hip.i3(lo), _hip.i3(hi), sums.data_ptr(), _stream(a)),
                   "m3d_box_corr_sums")
    return sums.cpu().numpy()


def separable_filter(volume: torch.Tensor, weights, offsets, wrap: bool,
                     out: torch.Tensor | None = None) -> torch.Tensor:
    """Per-axis FIR (``weights[a]`` = 1-D array of odd length or None to skip the axis)."""
    import ctypes

    volume = volume.contiguous()
    if out is None:
        out = torch.empty_like(volume)
    scratch = torch.empty_like(volume)
    keep = []
    ptrs = (ctypes.c_void_p * 3)()
    radius = []
    for a in range(3):
        w = weights[a]
        if w is None:
            radius.append(-1)
            ptrs[a] = None
            continue
        arr = np.ascontiguousarray(w, dtype=np.float32)
        if arr.size % 2 != 1:
            raise ValueError("filter length must be odd")
        keep.append(arr)
        ptrs[a] = arr.ctypes.data
        radius.append((arr.size - 1) // 2)
    with torch.cuda.device(volume.device):
        _hip.check(_hip.lib().m3d_separable_filter(volume.data_ptr(), out.data_ptr(), scratch.data_ptr(),
                                                   _hip.i3(volume.shape), ptrs, _hip.i3(radius),
                                                   _hip.i3(offsets), int(bool(wrap)), _stream(volume)),
                   "m3d_separable_filter")
    return out


# ----------------------------------------------------------------------------
# residual flow estimation (utils/sofima_registration.py)
# ----------------------------------------------------------------------------
def _i3(v):
    import ctypes

    return (ctypes.c_int32 * 3)(*[int(x) for x in v])


def _f3(v):
    import ctypes

    return (ctypes.c_float * 3)(*[float(x) for x in v])


def patch_pack(pre: torch.Tensor, post: torch.Tensor, patch_zyx, step_zyx, grid_zyx, node_begin: int,
               node_count: int, fft_zyx, out: torch.Tensor) -> torch.Tensor:
    """Mean-subtracted pre + i*post patches of ``node_count`` flow-grid nodes, zero padded."""
    if pre.dtype != torch.float32 or post.dtype != torch.float32 or pre.shape != post.shape or pre.ndim != 3:
        raise ValueError("patch_pack expects two float32 volumes of the same 3-D shape")
    if out.dtype != torch.complex64 or not out.is_contiguous() or \
            out.numel() < int(node_count) * int(np.prod([int(v) for v in fft_zyx])):
        raise ValueError("patch_pack output must be contiguous complex64 [count, fz, fy, fx]")
    pre, post = pre.contiguous(), post.contiguous()
    with torch.cuda.device(pre.device):
        _hip.check(_hip.lib().m3d_patch_pack(pre.data_ptr(), post.data_ptr(), _i3(pre.shape), _i3(patch_zyx),
                                             _i3(step_zyx), _i3(grid_zyx), int(node_begin), int(node_count),
                                             _i3(fft_zyx), out.data_ptr(), _stream(pre)), "m3d_patch_pack")
    return out


def patch_cross_spectrum(buf: torch.Tensor, count: int, fft_zyx) -> torch.Tensor:
    with torch.cuda.device(buf.device):
        _hip.check(_hip.lib().m3d_patch_cross_spectrum(buf.data_ptr(), int(count), _i3(fft_zyx), _stream(buf)),
                   "m3d_patch_cross_spectrum")
    return buf


def xcorr_peaks(buf: torch.Tensor, count: int, fft_zyx, patch_zyx, min_distance: int, peak_radius: int,
                threshold_rel: float, flow_out: torch.Tensor, node_begin: int) -> torch.Tensor:
    """``flow_out`` float32 [5, nodes]: x, y, z, sharpness, ratio of nodes [node_begin, +count)."""
    if flow_out.dtype != torch.float32 or flow_out.ndim != 2 or flow_out.shape[0] != 5 or \
            not flow_out.is_contiguous():
        raise ValueError("flow_out must be contiguous float32 [5, nodes]")
    with torch.cuda.device(buf.device):
        _hip.check(_hip.lib().m3d_xcorr_peaks(buf.data_ptr(), int(count), _i3(fft_zyx), _i3(patch_zyx),
                                              int(min_distance), int(peak_radius), float(threshold_rel),
                                              flow_out.data_ptr(), int(flow_out.shape[1]), int(node_begin),
                                              _stream(buf)), "m3d_xcorr_peaks")
    return flow_out


def patch_xcorr_fused_supported(patch_zyx, fft_zyx) -> bool:
    """True when the fused cross-correlation kernels cover this patch / transform box (24 x 64 x 64)."""
    return bool(_hip.lib().m3d_patch_xcorr_fused_supported(_i3(patch_zyx), _i3(fft_zyx)))


def patch_xcorr_fused_workspace(patch_zyx, count: int, device) -> torch.Tensor:
    nbytes = int(_hip.lib().m3d_patch_xcorr_fused_workspace_bytes(_i3(patch_zyx), int(count)))
    return torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)


def patch_xcorr_fused(pre: torch.Tensor, post: torch.Tensor, patch_zyx, step_zyx, grid_zyx, node_begin: int,
                      node_count: int, min_distance: int, peak_radius: int, threshold_rel: float,
                      work: torch.Tensor, flow_out: torch.Tensor) -> torch.Tensor:
    """patch_pack + forward transform + patch_cross_spectrum + inverse transform + xcorr_peaks for ``node_count``
    nodes in one call, transforms in LDS (``flow_out`` float32 [5, nodes] as for :func:`xcorr_peaks`)."""
    if pre.dtype != torch.float32 or post.dtype != torch.float32 or pre.shape != post.shape or pre.ndim != 3:
        raise ValueError("patch_xcorr_fused expects two float32 volumes of the same 3-D shape")
    if flow_out.dtype != torch.float32 or flow_out.ndim != 2 or flow_out.shape[0] != 5 or \
            not flow_out.is_contiguous():
        raise ValueError("flow_out must be contiguous float32 [5, nodes]")
    pre, post = pre.contiguous(), post.contiguous()
    with torch.cuda.device(pre.device):
        _hip.check(_hip.lib().m3d_patch_xcorr_fused(
            pre.data_ptr(), post.data_ptr(), _i3(pre.shape), _i3(patch_zyx), _i3(step_zyx), _i3(grid_zyx),
            int(node_begin), int(node_count), int(min_distance), int(peak_radius), float(threshold_rel),
            work.data_ptr(), int(work.numel()), flow_out.data_ptr(), int(flow_out.shape[1]), _stream(pre)),
            "m3d_patch_xcorr_fused")
    return flow_out


def subpixel_refine(pre: torch.Tensor, post: torch.Tensor, patch_zyx, step_zyx, nodes_zyx: torch.Tensor,
                    vectors_xyz: torch.Tensor, offsets, epsilon: float) -> torch.Tensor:
    import ctypes

    n = int(nodes_zyx.shape[0])
    nodes_zyx = nodes_zyx.to(torch.int32).contiguous()
    vectors_xyz = vectors_xyz.to(torch.float32).contiguous()
    out = torch.empty_like(vectors_xyz)
    offs = (ctypes.c_float * len(offsets))(*[float(v) for v in offsets])
    pre, post = pre.contiguous(), post.contiguous()
    with torch.cuda.device(pre.device):
        _hip.check(_hip.lib().m3d_subpixel_refine(pre.data_ptr(), post.data_ptr(), _i3(pre.shape),
                                                  _i3(patch_zyx), _i3(step_zyx), nodes_zyx.data_ptr(),
                                                  vectors_xyz.data_ptr(), n, offs, len(offsets), float(epsilon),
                                                  out.data_ptr(), _stream(pre)), "m3d_subpixel_refine")
    return out


def mesh_relax(x_xyz: torch.Tensor, prev_xyz: torch.Tensor, stride_xyz, *, dt, k0, k, stop_v_max, dt_max,
               start_cap, final_cap, num_iters, max_iters, cap_upscale=10.0):
    """Relax ``x_xyz`` [3, gz, gy, gx] in place; returns (x, MeshResult)."""
    import ctypes

    if x_xyz.dtype != torch.float32 or x_xyz.ndim != 4 or x_xyz.shape[0] != 3 or x_xyz.shape != prev_xyz.shape:
        raise ValueError("mesh_relax expects float32 [3, gz, gy, gx] positions and measurements")
    if not x_xyz.is_contiguous():
        raise ValueError("mesh_relax positions must be contiguous (relaxed in place)")
    prev_xyz = prev_xyz.to(torch.float32).contiguous()
    grid = _i3(x_xyz.shape[1:])
    nbytes = int(_hip.lib().m3d_mesh_relax_workspace_bytes(grid))
    work = torch.empty(nbytes, dtype=torch.uint8, device=x_xyz.device)
    params = _hip.MeshParams(float(dt), float(k0), float(k), float(stop_v_max), float(dt_max), float(start_cap),
                             float(final_cap), float(cap_upscale), int(num_iters), int(max_iters))
    result = _hip.MeshResult()
    with torch.cuda.device(x_xyz.device):
        _hip.check(_hip.lib().m3d_mesh_relax(x_xyz.data_ptr(), prev_xyz.data_ptr(), grid, _f3(stride_xyz),
                                             ctypes.byref(params), work.data_ptr(), nbytes, ctypes.byref(result),
                                             _stream(x_xyz)), "m3d_mesh_relax")
    return x_xyz, result


def flow_compose(base: torch.Tensor, residual: torch.Tensor, stride_zyx, box_start_xyz) -> torch.Tensor:
    base = base.to(torch.float32).contiguous()
    residual = residual.to(torch.float32).contiguous()
    if base.shape != residual.shape or base.ndim != 4 or base.shape[0] != 3:
        raise ValueError("flow_compose expects two [3, gz, gy, gx] fields on the same grid")
    out = torch.empty_like(base)
    with torch.cuda.device(base.device):
        _hip.check(_hip.lib().m3d_flow_compose(base.data_ptr(), residual.data_ptr(), _i3(base.shape[1:]),
                                               _f3(stride_zyx), _f3(box_start_xyz), out.data_ptr(),
                                               _stream(base)), "m3d_flow_compose")
    return out


def moments(volume: torch.Tensor) -> tuple[float, float]:
    """(mean, population std) of a float32 tensor, double accumulation."""
    volume = volume.contiguous()
    if volume.dtype != torch.float32:
        raise ValueError("moments expects float32")
    out = torch.empty(2, dtype=torch.float64, device=volume.device)
    with torch.cuda.device(volume.device):
        _hip.check(_hip.lib().m3d_moments(volume.data_ptr(), volume.numel(), out.data_ptr(), _stream(volume)),
                   "m3d_moments")
    s, q = out.tolist()
    n = max(volume.numel(), 1)
    mean = s / n
    return mean, float(np.sqrt(max(q / n - mean * mean, 0.0)))


def normalized_sq_diff_mean(a: torch.Tensor, b: torch.Tensor, mean_a, scale_a, mean_b, scale_b) -> float:
    a, b = a.contiguous(), b.contiguous()
    if a.dtype != torch.float32 or b.dtype != torch.float32 or a.shape != b.shape:
        raise ValueError("normalized_sq_diff_mean expects two float32 tensors of one shape")
    out = torch.empty(1, dtype=torch.float64, device=a.device)
    with torch.cuda.device(a.device):
        _hip.check(_hip.lib().m3d_normalized_sq_diff(a.data_ptr(), b.data_ptr(), a.numel(), float(mean_a),
                                                     float(scale_a), float(mean_b), float(scale_b),
                                                     out.data_ptr(), _stream(a)), "m3d_normalized_sq_diff")
    return float(out.item()) / max(a.numel(), 1)
