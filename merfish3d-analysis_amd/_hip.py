"""ctypes binding of ``csrc/libm3d_hip.so`` - the C ABI declared in ``include/m3d_hip.h``.

PyTorch-ROCm supplies device memory (``tensor.data_ptr()``) and streams
(``torch.cuda.current_stream().cuda_stream``); every kernel on the hot path
lives in the shared library.  There is no CPU fallback: if the library is
missing this module raises at first use.
"""

from __future__ import annotations

import ctypes
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent / "csrc" / "libm3d_hip.so"
_lib: ctypes.CDLL | None = None

M3D_U16, M3D_F32, M3D_F16, M3D_I16 = 0, 1, 2, 3
TRACE_CAP = 512
ENGINE_X_STATIC, ENGINE_Y_STATIC, ENGINE_SPLIT_CHAIN, ENGINE_UPDATE_CHAIN, ENGINE_Z_REGISTER = 1, 2, 4, 8, 16


class M3DError(RuntimeError):
    """Non-zero status from libm3d_hip.so that has no more specific Python type."""


class RLGCParams(ctypes.Structure):
    _fields_ = [
        ("safe_mode", c_int32),
        ("limit", c_float),
        ("max_delta", c_float),
        ("seed", c_uint64),
        ("normalize_psf", c_int32),
        ("max_iterations", c_int32),
        ("profile", c_int32),
        ("reserved", c_int32),
    ]


class MeshParams(ctypes.Structure):
    _fields_ = [
        ("dt", c_float), ("k0", c_float), ("k", c_float), ("stop_v_max", c_float), ("dt_max", c_float),
        ("start_cap", c_float), ("final_cap", c_float), ("cap_upscale", c_float),
        ("num_iters", c_int32), ("max_iters", c_int32),
    ]


class MeshResult(ctypes.Structure):
    _fields_ = [
        ("iterations", c_int32), ("blocks", c_int32), ("final_kinetic_energy", ctypes.c_double),
        ("final_v_max", c_float), ("final_cap", c_float), ("final_dt", c_float), ("reserved", c_float),
    ]


class RLGCResult(ctypes.Structure):
    _fields_ = [
        ("iterations", c_int32),
        ("stop_reason", c_int32),
        ("n_trace", c_int32),
        ("engine_flags", c_int32),
        ("kld_image", ctypes.c_double * TRACE_CAP),
        ("kld_split1", ctypes.c_double * TRACE_CAP),
        ("kld_split2", ctypes.c_double * TRACE_CAP),
        ("updated_fraction", ctypes.c_double * TRACE_CAP),
        ("max_relative_delta", ctypes.c_double * TRACE_CAP),
        ("update_min", ctypes.c_double * TRACE_CAP),
        ("update_max", ctypes.c_double * TRACE_CAP),
        ("conv_ms_total", ctypes.c_double),
        ("conv_calls", c_int32),
        ("fused_engine", c_int32),
        ("kernel_ms", ctypes.c_double * 5),
        ("loop_ms", ctypes.c_double),
    ]


class BloscHeader(ctypes.Structure):
    _fields_ = [
        ("nbytes", ctypes.c_uint32), ("blocksize", ctypes.c_uint32), ("cbytes", ctypes.c_uint32),
        ("typesize", c_int32), ("shuffle", c_int32), ("codec", c_int32), ("memcpyed", c_int32),
        ("split", c_int32), ("nblocks", c_int32),
    ]


class IOStats(ctypes.Structure):
    _fields_ = [
        ("seconds_total", ctypes.c_double), ("seconds_entropy", ctypes.c_double),
        ("file_bytes", c_int64), ("raw_bytes", c_int64), ("chunks", c_int32), ("chunks_missing", c_int32),
    ]


STOP_REASONS = {0: "", 1: "restore_previous_recon", 2: "limit", 3: "max_delta", 4: "max_iterations"}

_I3 = c_int32 * 3
_F3 = c_float * 3
_L3 = c_int64 * 3

_SIGNATURES = {
    "m3d_version": (c_int, []),
    "m3d_source_hash": (c_char_p, []),
    "m3d_status_string": (c_char_p, [c_int]),
    "m3d_last_error": (c_char_p, []),
    "m3d_next_fft_size": (c_int, [c_int]),
    "m3d_rlgc_geometry": (c_int, [_I3, _I3, c_int, _I3, _I3, _I3]),
    "m3d_set_fft_engine": (c_int, [c_int]),
    "m3d_rlgc_plan_uses_rocfft": (c_int, [c_void_p]),
    "m3d_rlgc_plan_create": (c_int, [_I3, _I3, c_int, POINTER(c_void_p)]),
    "m3d_rlgc_plan_workspace_bytes": (c_size_t, [c_void_p]),
    "m3d_rlgc_plan_destroy": (c_int, [c_void_p]),
    "m3d_rlgc_run": (
        c_int,
        [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
         POINTER(RLGCParams), c_void_p, c_int, POINTER(RLGCResult), c_void_p],
    ),
    "m3d_rlgc_debug_split": (c_int, [c_void_p, c_int, c_int64, c_uint64, c_int32, c_void_p, c_void_p]),
    "m3d_binom_tables_host": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "m3d_binom_debug_table_draw": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "m3d_fft_debug_pk": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "m3d_clip_to_u16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "m3d_weight_readout": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "m3d_gaussian_lowpass": (c_int, [c_void_p, c_void_p, c_void_p, _I3, _F3, c_int, c_void_p]),
    "m3d_decode_pixels": (
        c_int,
        [c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_int32, c_float, c_float,
         c_float, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
         c_int32, c_void_p],
    ),
    "m3d_ccl_label": (c_int, [c_void_p, _I3, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "m3d_region_stats": (
        c_int,
        [c_void_p, c_void_p, c_void_p, _I3, c_int32, c_void_p, c_int, c_void_p, c_void_p, c_int32,
         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "m3d_radix_hist": (
        c_int,
        [c_void_p, c_int64, c_int, c_float, c_int, ctypes.c_double, c_int, ctypes.c_uint32, c_void_p, c_void_p],
    ),
    "m3d_replace_above": (c_int, [c_void_p, c_int64, c_float, c_float, c_void_p]),
    "m3d_separable_filter": (
        c_int,
        [c_void_p, c_void_p, c_void_p, _I3, POINTER(c_void_p), _I3, _I3, c_int, c_void_p],
    ),
    "m3d_fft_plan_create": (c_int, [c_int, POINTER(c_int32), POINTER(c_void_p)]),
    "m3d_fft_plan_workspace_bytes": (c_size_t, [c_void_p]),
    "m3d_fft_exec": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "m3d_fft_plan_destroy": (c_int, [c_void_p]),
    "m3d_real_to_complex": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "m3d_cross_power": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "m3d_argmax_abs": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "m3d_contract_last": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int32, c_int, c_void_p, c_void_p]),
    "m3d_box_corr_sums": (c_int, [c_void_p, c_void_p, _I3, _I3, _I3, c_void_p, c_void_p]),
    "m3d_max_z": (c_int, [c_void_p, c_int, _I3, c_void_p, c_void_p]),
    "m3d_affine_warp": (
        c_int,
        [c_void_p, c_int, _I3, ctypes.c_double * 9, ctypes.c_double * 3, c_void_p, _I3, c_int,
         ctypes.c_double, c_void_p],
    ),
    "m3d_flow_warp": (
        c_int,
        [c_void_p, c_int, _I3, c_void_p, _I3, c_void_p, _I3, _F3, _F3, _F3, _F3, c_float * 16,
         c_int, c_float, c_void_p],
    ),
    "m3d_fft_plan_create_batched": (c_int, [c_int, POINTER(c_int32), c_int32, POINTER(c_void_p)]),
    "m3d_patch_pack": (c_int, [c_void_p, c_void_p, _I3, _I3, _I3, _I3, c_int64, c_int32, _I3, c_void_p, c_void_p]),
    "m3d_patch_cross_spectrum": (c_int, [c_void_p, c_int32, _I3, c_void_p]),
    "m3d_xcorr_peaks": (
        c_int,
        [c_void_p, c_int32, _I3, _I3, c_int32, c_int32, c_float, c_void_p, c_int64, c_int64, c_void_p],
    ),
    "m3d_patch_xcorr_fused_supported": (c_int, [_I3, _I3]),
    "m3d_patch_xcorr_fused_workspace_bytes": (c_size_t, [_I3, c_int32]),
    "m3d_patch_xcorr_fused": (
        c_int,
        [c_void_p, c_void_p, _I3, _I3, _I3, _I3, c_int64, c_int32, c_int32, c_int32, c_float, c_void_p, c_size_t,
         c_void_p, c_int64, c_void_p],
    ),
    "m3d_subpixel_refine": (
        c_int,
        [c_void_p, c_void_p, _I3, _I3, _I3, c_void_p, c_void_p, c_int32, POINTER(c_float), c_int32, c_float,
         c_void_p, c_void_p],
    ),
    "m3d_mesh_relax_workspace_bytes": (c_size_t, [_I3]),
    "m3d_mesh_relax": (
        c_int,
        [c_void_p, c_void_p, _I3, _F3, POINTER(MeshParams), c_void_p, c_size_t, POINTER(MeshResult), c_void_p],
    ),
    "m3d_flow_compose": (c_int, [c_void_p, c_void_p, _I3, _F3, _F3, c_void_p, c_void_p]),
    "m3d_moments": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "m3d_normalized_sq_diff": (
        c_int, [c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_void_p, c_void_p]),
    "m3d_filter_update": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "m3d_enforce_symmetric_boundary": (c_int, [c_void_p, _I3, _I3, _I3, c_void_p]),
    "m3d_observed_region_mask": (c_int, [c_void_p, _I3, _I3, _I3, c_void_p]),
    "m3d_pad_psf_workspace_bytes": (c_size_t, []),
    "m3d_pad_psf": (c_int, [c_void_p, _I3, c_void_p, _I3, c_int, c_void_p, c_size_t, c_void_p]),
    "m3d_kl_div_workspace_bytes": (c_size_t, []),
    "m3d_kl_div": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_size_t, POINTER(ctypes.c_double),
                           c_void_p]),
    "m3d_fftconv_create": (c_int, [_I3, POINTER(c_void_p)]),
    "m3d_fftconv_destroy": (c_int, [c_void_p]),
    "m3d_fftconv_spectrum_bytes": (c_size_t, [c_void_p]),
    "m3d_fftconv_otf": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "m3d_fftconv_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    # include/m3d_io.h
    "m3d_blosc_info": (c_int, [c_void_p, c_size_t, POINTER(BloscHeader)]),
    "m3d_blosc_decode_host": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t]),
    "m3d_blosc_bound": (c_size_t, [c_size_t, c_size_t]),
    "m3d_blosc_encode_host": (
        c_int, [c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_size_t, c_void_p, c_size_t, POINTER(c_size_t)]),
    "m3d_io_create": (c_int, [c_int, POINTER(c_void_p)]),
    "m3d_io_destroy": (None, [c_void_p]),
    "m3d_zarr_read_device": (c_int, [c_void_p, c_char_p, _L3, _L3, c_int, c_void_p, c_void_p, POINTER(IOStats)]),
    "m3d_zarr_read_host": (c_int, [c_void_p, c_char_p, _L3, _L3, c_int, c_void_p, POINTER(IOStats)]),
    "m3d_zarr_write_device": (
        c_int, [c_void_p, c_char_p, _L3, _L3, c_int, c_void_p, c_int, c_int, c_int, c_void_p, POINTER(IOStats)]),
    "m3d_zarr_write_host": (
        c_int, [c_void_p, c_char_p, _L3, _L3, c_int, c_void_p, c_int, c_int, c_int, POINTER(IOStats)]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def library_path() -> Path:
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    """Load the HIP library (after torch, so both share one HIP runtime / rocFFT)."""
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise ImportError(
                f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback."
            )
        import torch  # noqa: F401  (loads libamdhip64 / librocfft first; same SONAMEs are reused)

        handle = ctypes.CDLL(str(_LIB_PATH), mode=ctypes.RTLD_GLOBAL)
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(handle, name, None)
            if fn is None:
                raise ImportError(f"{_LIB_PATH} lacks {name}: rebuild it (python merfish3d-analysis_amd/csrc/build.py)")
            fn.restype = restype
            fn.argtypes = argtypes
        _refuse_stale_library(handle)
        _lib = handle
    return _lib


def _refuse_stale_library(handle) -> None:
    """A library built from other sources than the tree's is not the product (and, half-rebuilt, can abort the
    process inside the HIP runtime - DESIGN.md section 11).  Checked where the sources travel with the library."""
    from . import _buildcheck

    if not any(_buildcheck.CSRC.glob("*.hip")):
        return  # binary-only install: nothing to compare with
    built, tree = handle.m3d_source_hash().decode(), _buildcheck.source_hash()
    if built != tree:
        raise ImportError(
            f"{_LIB_PATH} was built from sources {built[:16]}, the tree holds {tree[:16]}: rebuild it "
            "(python merfish3d-analysis_amd/csrc/build.py).  There is no CPU fallback."
        )


def check(status: int, what: str = "") -> None:
    """Map the C status codes to the reference's exception types."""
    if status == 0:
        return
    handle = lib()
    detail = handle.m3d_last_error().decode(errors="replace")
    text = handle.m3d_status_string(status).decode()
    msg = f"{what + ': ' if what else ''}{text} ({detail})"
    if status in (-1, -5):
        raise ValueError(msg)
    if status == -2:
        raise MemoryError(msg + " - out of memory")
    if status == -6:
        raise NotImplementedError(msg)
    raise M3DError(msg)


def i3(values) -> ctypes.Array:
    return _I3(*[int(v) for v in values])


def l3(values) -> ctypes.Array:
    return _L3(*[int(v) for v in values])


def f3(values) -> ctypes.Array:
    return _F3(*[float(v) for v in values])


def ptr(tensor) -> int | None:
    """Device pointer of a contiguous CUDA(HIP) torch tensor (None passes through)."""
    if tensor is None:
        return None
    if not tensor.is_cuda:
        raise ValueError("expected a tensor in GPU memory")
    if not tensor.is_contiguous():
        raise ValueError("expected a contiguous tensor")
    return tensor.data_ptr()


def current_stream(device=None) -> int:
    import torch

    return torch.cuda.current_stream(device).cuda_stream
