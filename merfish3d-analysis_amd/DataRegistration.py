"""Per-tile preprocess operator on MI355X (mirror of the local part of DataRegistration.py).

Same constructor, properties and entry points as the reference
(``DataRegistration.py:1156-1232, 1522-1591, 2086-2315``):

    register_all_tiles / register_one_tile / apply_registration_to_one_tile

For one tile: deconvolve the round-1 fiducial (RLGC, PSF 0) and store an identity transform;
for every other round deconvolve, register to round 1 (``register_pair_to_fixed``) and store
the 4 x 4 physical transform; optionally estimate the residual SOFIMA flow field of each
moving round; then for every readout bit optionally deconvolve (PSF 1 for excitation
< 0.600 um, else PSF 2), run the feature predictor and store its image and spot table.

What differs from the reference, by design:

* One process per GPU.  With ``num_gpus == 1``, or under ``torch.distributed`` (whole tiles are
  dealt to the ranks first; only with fewer tiles than ranks does every rank take a contiguous
  share of one tile's rounds / bits, the reference's :2137-2141 split), the work runs in this
  process and volumes stay in HBM between deconvolution, clipping and registration - the
  reference pickles the uint16 reference volume into every worker and re-uploads it.
  ``num_gpus > 1`` without a process group spawns one child per GPU like the reference
  (``HIP_VISIBLE_DEVICES`` pins the device).
* The feature predictor (U-FISH, a third-party ONNX network, :1012-1017) is outside the hot
  path: pass ``feature_predictor=callable(image_zyx_uint16) -> (spots DataFrame with
  axis-0/1/2 columns, float32 image)``, or install ``ufish`` and the reference's call is made.
* Global (multiview-stitcher) registration is outside the hot path: ``global_registration=True``
  raises ``NotImplementedError`` when reached.
"""

from __future__ import annotations

import gc
import queue
import timeit
import traceback
from collections.abc import Callable
from datetime import datetime
from pathlib import Path
from typing import Any

import numpy as np

from .utils.sofima_registration import SofimaRegistrationConfig

_REGISTRATION_DIAG_PREFIX = "[registration-diag]"


def _registration_diag(message: str, *, enabled: bool = False) -> None:
    if enabled:
        print(f"{_REGISTRATION_DIAG_PREFIX} {message}", flush=True)


def _restrict_worker_to_assigned_gpu(gpu_id: int) -> int:
    """Restrict a spawned worker process to one physical GPU before anything initialises HIP
    (DataRegistration.py:151-167: ``CUDA_VISIBLE_DEVICES`` there, ``HIP_VISIBLE_DEVICES`` on ROCm);
    the device is index 0 inside the worker afterwards."""
    import os

    os.environ["HIP_VISIBLE_DEVICES"] = str(int(gpu_id))
    return 0


def time_stamp() -> str:
    """``YYYY-MM-DD HH:MM:SS`` (:2318-2327)."""
    return datetime.now().strftime("%Y-%m-%d %H:%M:%S")


def _result(value):
    return value.result() if hasattr(value, "result") else value


def _resolve_psf(psfs: Any, psf_idx: int) -> np.ndarray:
    """PSF by channel index from a list or a stacked array (:235-264)."""
    if isinstance(psfs, list):
        if psf_idx < 0 or psf_idx >= len(psfs):
            raise IndexError(f"PSF index {psf_idx} out of range for {len(psfs)} PSFs.")
        return np.asarray(psfs[psf_idx], dtype=np.float32)
    psf_array = np.asarray(psfs)
    if psf_array.ndim < 3:
        raise ValueError(f"Invalid PSF array shape: {psf_array.shape}")
    if psf_idx < 0 or psf_idx >= psf_array.shape[0]:
        raise IndexError(f"PSF index {psf_idx} out of range for shape {psf_array.shape}.")
    return np.asarray(psf_array[psf_idx], dtype=np.float32)


def _release_worker_gpu_memory(clear_plans: bool = False) -> None:
    """Return cached HBM to the driver between stages (:267-304)."""
    import torch

    from .utils.rlgc import clear_rlgc_caches

    gc.collect()
    if clear_plans:
        clear_rlgc_caches(clear_memory_pool=True)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


def _run_chunked_rlgc_remembering_crop(dr: Any, chunked_rlgc: Any, image, psf: np.ndarray, gpu_id: int,
                                       release_memory: bool = False):
    """RLGC that remembers the lateral chunk size that fitted (:307-387)."""
    shape = tuple(int(v) for v in image.shape)
    image_max_yx = max(shape) if len(shape) == 2 else max(shape[-2:])

    def _remember_successful_crop(successful_crop_yx: int) -> None:
        previous_crop_yx = int(dr._crop_yx_decon)
        requested_crop_yx = min(previous_crop_yx, int(image_max_yx))
        if int(successful_crop_yx) >= requested_crop_yx:
            return
        dr._crop_yx_decon = int(successful_crop_yx)
        if dr._verbose >= 1:
            print(time_stamp(), "RLGC reduced crop_yx after GPU memory fallback: "
                                f"{previous_crop_yx} -> {successful_crop_yx}.")

    return chunked_rlgc(image=image, psf=psf, gpu_id=gpu_id, crop_yx=dr._crop_yx_decon, crop_z=None,
                        release_memory=release_memory, on_successful_crop_yx=_remember_successful_crop)


def _to_device_u16(array, gpu_id: int):
    import torch

    if isinstance(array, torch.Tensor):
        return array.to(f"cuda:{gpu_id}")
    arr = np.ascontiguousarray(array)
    if arr.dtype != np.uint16:
        arr = arr.astype(np.uint16)
    return torch.from_numpy(arr).to(f"cuda:{gpu_id}")


def _deconvolve_to_u16(dr, raw, psf_idx: int, gpu_id: int):
    """RLGC then ``clip(0, 65535).astype(uint16)`` (:427-435, :985-995) with the volume staying
    in HBM; returns a uint16 tensor."""
    from . import kernels
    from .utils.rlgc import chunked_rlgc

    decon = _run_chunked_rlgc_remembering_crop(dr=dr, chunked_rlgc=chunked_rlgc, image=_to_device_u16(raw, gpu_id),
                                               psf=_resolve_psf(dr._psfs, psf_idx), gpu_id=gpu_id,
                                               release_memory=True)
    import torch

    if not isinstance(decon, torch.Tensor):
        decon = torch.from_numpy(np.ascontiguousarray(decon, dtype=np.float32)).to(f"cuda:{gpu_id}")
    return kernels.clip_to_u16(decon)


def _load_deconvolve_fiducial_round(dr, round_id: str, gpu_id: int, chunked_rlgc: Any = None):
    """Load and optionally deconvolve one fiducial round -> uint16 tensor in HBM (:390-453)."""
    raw = _result(dr._datastore.load_local_corrected_image(tile=dr._tile_id, round=round_id, return_future=False))
    _registration_diag(f"fiducial_decon_start tile={dr._tile_id} round={round_id} "
                       f"raw_shape={tuple(int(v) for v in raw.shape)}", enabled=dr._registration_diagnostics)
    start_time = timeit.default_timer()
    if dr._decon_fiducial:
        decon = _deconvolve_to_u16(dr, raw, 0, gpu_id)
        dr._datastore.save_local_deconvolved_fiducial_image(decon.cpu().numpy(), tile=dr._tile_id, round=round_id)
    else:
        decon = _to_device_u16(np.asarray(raw).astype(np.uint16), gpu_id)
    _registration_diag(f"fiducial_decon_done tile={dr._tile_id} round={round_id} "
                       f"output_shape={tuple(int(v) for v in decon.shape)} "
                       f"elapsed_s={timeit.default_timer() - start_time:.2f}", enabled=dr._registration_diagnostics)
    return decon


def _process_fiducial_rounds_on_gpu(dr, round_list: list, reference_image, gpu_id: int, result_queue: Any) -> None:
    """Deconvolve, register and store the transform of each assigned moving round (:456-558)."""
    import torch

    from .utils.multiview_registration import register_pair_to_fixed

    try:
        torch.cuda.set_device(gpu_id)
        spacing_zyx_um = dr._datastore.voxel_size_zyx_um
        reference = _to_device_u16(reference_image, gpu_id)
        for round_id in round_list:
            decon = _load_deconvolve_fiducial_round(dr=dr, round_id=round_id, gpu_id=gpu_id)
            _release_worker_gpu_memory()
            start_time = timeit.default_timer()
            local_transform_zyx_um = register_pair_to_fixed(reference, decon, spacing_zyx_um=spacing_zyx_um,
                                                            diagnostics=dr._registration_diagnostics)
            dr._datastore.save_local_round_transform_zyx_um(
                transform_zyx_um=np.asarray(local_transform_zyx_um, dtype=np.float32), tile=dr._tile_id,
                round=round_id)
            _registration_diag(f"fiducial_phase_registration_done tile={dr._tile_id} round={round_id} "
                               f"elapsed_s={timeit.default_timer() - start_time:.2f}",
                               enabled=dr._registration_diagnostics)
            if dr._verbose >= 1:
                print(time_stamp(), f"Finished fiducial tile id: {dr._tile_id}; round id: {round_id}.")
            result_queue.put(("result", round_id, int(dr._crop_yx_decon)))
            del decon
            _release_worker_gpu_memory()
    except Exception:
        result_queue.put(("error", None, traceback.format_exc()))
        raise


def _process_sofima_rounds_on_gpu(dr, round_list: list, gpu_id: int, result_queue: Any) -> None:
    """Affine-warp each moving fiducial onto round 1 and store its residual flow field (:561-693)."""
    import torch

    from .utils.multiview_registration import warp_array_to_reference_gpu
    from .utils.sofima_registration import estimate_sofima_flow_field_xyz_px

    try:
        torch.cuda.set_device(gpu_id)
        spacing_zyx_um = dr._datastore.voxel_size_zyx_um
        reference_round_id = dr._round_ids[0]
        fixed = _result(dr._datastore.load_local_fiducial_image(tile=dr._tile_id, round=reference_round_id,
                                                                return_future=False))
        if fixed is None:
            raise RuntimeError(f"Missing reference fiducial for tile={dr._tile_id} round={reference_round_id}.")
        fixed = _to_device_u16(fixed, gpu_id).to(torch.float32)
        for round_id in round_list:
            moving = _result(dr._datastore.load_local_fiducial_image(tile=dr._tile_id, round=round_id,
                                                                     return_future=False))
            if moving is None:
                raise RuntimeError(f"Missing moving fiducial for tile={dr._tile_id} round={round_id}.")
            transform_zyx_um = dr._datastore.load_local_round_transform_zyx_um(tile=dr._tile_id, round=round_id)
            if transform_zyx_um is None:
                raise RuntimeError(f"Missing local affine transform for tile={dr._tile_id} round={round_id}.")
            warped = warp_array_to_reference_gpu(_to_device_u16(moving, gpu_id), transform_zyx_um=transform_zyx_um,
                                                 spacing_zyx_um=spacing_zyx_um, reference_shape=tuple(fixed.shape),
                                                 gpu_id=gpu_id, diagnostics=dr._registration_diagnostics)
            sofima_start_time = timeit.default_timer()
            sofima_flow_field, metadata = estimate_sofima_flow_field_xyz_px(fixed, warped.to(torch.float32),
                                                                            config=dr._sofima_config)
            dr._datastore.save_local_sofima_flow_field(
                sofima_flow_field, tile=dr._tile_id, round=round_id, reference_round=reference_round_id,
                map_stride_zyx_px=metadata["map_stride_zyx_px"],
                map_box_start_xyz_px=metadata["map_box_start_xyz_px"],
                map_box_size_xyz_px=metadata["map_box_size_xyz_px"],
                reference_shape_zyx_px=tuple(int(v) for v in fixed.shape),
                moving_shape_zyx_px=tuple(int(v) for v in warped.shape),
                sofima_status=metadata.get("status", "ok"),
                valid_flow_vectors=metadata.get("valid_flow_vectors"), return_future=False)
            _registration_diag(f"fiducial_sofima_done tile={dr._tile_id} round={round_id} "
                               f"status={metadata.get('status', 'ok')} "
                               f"valid_flow_vectors={metadata.get('valid_flow_vectors')} "
                               f"elapsed_s={timeit.default_timer() - sofima_start_time:.2f}",
                               enabled=dr._registration_diagnostics)
            result_queue.put(("result", round_id, None))
            del moving, warped, sofima_flow_field
            _release_worker_gpu_memory()
    except Exception:
        result_queue.put(("error", None, traceback.format_exc()))
        raise


def _roi_sums(spots, image: np.ndarray, roi_dims) -> np.ndarray:
    """Sum of ``image`` in the clipped ROI around every spot (``sum_pixels_in_roi``, :1049-1083):
    start = max(0, c - r // 2), stop = min(size, start + r)."""
    out = np.empty(len(spots), dtype=np.float64)
    rz, ry, rx = roi_dims
    for i, (z, y, x) in enumerate(zip(spots["z"].to_numpy(), spots["y"].to_numpy(), spots["x"].to_numpy())):
        zmin, ymin, xmin = max(0, z - rz // 2), max(0, y - ry // 2), max(0, x - rx // 2)
        zmax, ymax, xmax = min(image.shape[0], zmin + rz), min(image.shape[1], ymin + ry), min(image.shape[2],
                                                                                                xmin + rx)
        out[i] = np.sum(image[int(zmin):int(zmax), int(ymin):int(ymax), int(xmin):int(xmax)])
    return out


def _default_feature_predictor(dr, gpu_id: int):
    """The reference's U-FISH call (:1006-1017) when the package is installed."""
    try:
        from ufish.api import UFish
    except ImportError as exc:
        raise RuntimeError(
            "The feature predictor (U-FISH) is a third-party network outside this package: install `ufish` "
            "or pass feature_predictor=callable(image) -> (spots, image) to DataRegistration.") from exc
    ufish = UFish(device=f"cuda:{gpu_id}")
    ufish.load_weights() if dr._ufish_model is None else ufish.load_weights(weights_file=str(dr._ufish_model))

    def _predict(image):
        return ufish.predict(image, axes="zyx", blend_3d=False, batch_size=1)

    return _predict


def _apply_bits_on_gpu(dr, bit_list: list, gpu_id: int = 0) -> bool:
    """Optionally deconvolve each readout bit, run the feature predictor, save both (:906-1121)."""
    import torch

    torch.cuda.set_device(gpu_id)
    predictor = None
    for bit_id in bit_list:
        r_idx = dr._datastore.load_local_round_linker(tile=dr._tile_id, bit=bit_id) - 1
        ex_wl, _em_wl = dr._datastore.load_local_wavelengths_um(tile=dr._tile_id, bit=bit_id)
        psf_idx = 1 if ex_wl < 0.600 else 2
        decon_on_disk = dr._has_valid_deconvolved_readout_image(bit_id=bit_id)
        feature_predictor_on_disk = dr._has_valid_feature_predictor_outputs(bit_id=bit_id)
        if (decon_on_disk or not dr._decon_readout) and feature_predictor_on_disk and not dr._overwrite_outputs:
            continue
        corrected_image = _result(dr._datastore.load_local_corrected_image(tile=dr._tile_id, bit=bit_id,
                                                                           return_future=False))
        _registration_diag(f"bit_start tile={dr._tile_id} bit={bit_id} linked_round={dr._round_ids[r_idx]} "
                           f"raw_shape={tuple(int(v) for v in corrected_image.shape)}",
                           enabled=dr._registration_diagnostics)
        if dr._decon_readout:
            start_time = timeit.default_timer()
            predictor_input_image = _deconvolve_to_u16(dr, corrected_image, psf_idx, gpu_id).cpu().numpy()
            _registration_diag(f"bit_decon tile={dr._tile_id} bit={bit_id} "
                               f"elapsed_s={timeit.default_timer() - start_time:.2f}",
                               enabled=dr._registration_diagnostics)
            dr._datastore.save_local_deconvolved_readout_image(predictor_input_image, tile=dr._tile_id, bit=bit_id)
        else:
            predictor_input_image = np.asarray(corrected_image)
        if predictor is None:
            predictor = dr._feature_predictor or _default_feature_predictor(dr, gpu_id)
        feature_predictor_loc, feature_predictor_data = predictor(predictor_input_image)
        feature_predictor_loc = feature_predictor_loc.rename(columns={"axis-0": "z", "axis-1": "y", "axis-2": "x"})
        feature_predictor_data = np.asarray(feature_predictor_data)
        roi = (7, 5, 5)
        feature_predictor_loc["sum_prob_pixels"] = _roi_sums(feature_predictor_loc, feature_predictor_data, roi)
        feature_predictor_loc["sum_decon_pixels"] = _roi_sums(feature_predictor_loc, predictor_input_image, roi)
        feature_predictor_loc["tile_idx"] = dr._tile_ids.index(dr._tile_id)
        feature_predictor_loc["bit_idx"] = dr._bit_ids.index(bit_id) + 1
        feature_predictor_loc["tile_z_px"] = feature_predictor_loc["z"]
        feature_predictor_loc["tile_y_px"] = feature_predictor_loc["y"]
        feature_predictor_loc["tile_x_px"] = feature_predictor_loc["x"]
        dr._datastore.save_local_feature_predictor_image(feature_predictor_data, tile=dr._tile_id, bit=bit_id)
        dr._datastore.save_local_feature_predictor_spots(feature_predictor_loc, tile=dr._tile_id, bit=bit_id)
        if dr._verbose >= 1:
            print(time_stamp(), f"Finished readout tile id: {dr._tile_id}; bit id: {bit_id}.")
        del predictor_input_image, feature_predictor_data, feature_predictor_loc
        gc.collect()
    _release_worker_gpu_memory(clear_plans=False)
    return True


class _LocalQueue:
    """In-process stand-in for ``mp.Queue`` when the work runs in this process."""

    def __init__(self):
        self.items = []

    def put(self, item):
        self.items.append(item)


def _dist_ready() -> bool:
    try:
        import torch.distributed as dist

        return dist.is_available() and dist.is_initialized()
    except Exception:
        return False


class DataRegistration:
    """Register 2D or 3D MERFISH data across rounds (:1124-1232)."""

    def __init__(
        self,
        datastore,
        decon_fiducial: bool = True,
        decon_readout: bool = False,
        overwrite_outputs: bool = False,
        perform_deformable_registration: bool = True,
        num_gpus: int = 1,
        crop_yx_decon: int = 2048,
        ufish_model: str | Path | None = None,
        global_registration: bool = False,
        registration_diagnostics: bool = False,
        sofima_config: SofimaRegistrationConfig | None = None,
        global_registration_config: Any = None,
        verbose: int = 1,
        *,
        feature_predictor: Callable | None = None,
    ) -> None:
        self._datastore = datastore
        self._decon_fiducial = decon_fiducial
        self._tile_ids = self._datastore.tile_ids
        self._round_ids = self._datastore.round_ids
        self._bit_ids = self._datastore.bit_ids
        self._psfs = self._datastore.channel_psfs
        self._num_gpus = num_gpus
        self._crop_yx_decon = crop_yx_decon
        self._perform_deformable_registration = perform_deformable_registration
        self._data_raw = None
        self._overwrite_outputs = overwrite_outputs
        self._decon_readout = decon_readout
        self._ufish_model = ufish_model
        self._global_registration = global_registration
        self._registration_diagnostics = bool(registration_diagnostics)
        self._sofima_config = sofima_config or SofimaRegistrationConfig()
        self._global_registration_config = global_registration_config
        self._verbose = verbose
        self._feature_predictor = feature_predictor
        self._tile_id = None
        self._tile_parallel = False
        # rounds whose registration stands apart from their tile's other rounds (see _audit_round_transforms)
        self.suspect_registrations: dict = {}
        self.suspect_registrations_all_ranks: int | None = None

    # ----------------------------------------------------------------- properties (:1237-1344)
    @property
    def datastore(self):
        return self._datastore

    @datastore.setter
    def datastore(self, value) -> None:
        self._datastore = value

    @property
    def tile_id(self):
        if self._tile_id is not None:
            return self._tile_id
        print("Tile coordinate not defined.")
        return None

    @tile_id.setter
    def tile_id(self, value) -> None:
        if isinstance(value, (int, np.integer)):
            if value < 0 or value > self._datastore.num_tiles:
                print("Set value index >=0 and <=" + str(self._datastore.num_tiles))
                return None
            self._tile_id = self._datastore.tile_ids[int(value)]
        elif isinstance(value, str):
            if value not in self._datastore.tile_ids:
                print("set valid tile id")
                return None
            self._tile_id = value

    @property
    def perform_deformable_registration(self) -> bool:
        return self._perform_deformable_registration

    @perform_deformable_registration.setter
    def perform_deformable_registration(self, value: bool) -> None:
        self._perform_deformable_registration = value

    @property
    def overwrite_outputs(self) -> bool:
        return self._overwrite_outputs

    @overwrite_outputs.setter
    def overwrite_outputs(self, value: bool) -> None:
        self._overwrite_outputs = value

    # ----------------------------------------------------------------- resume helpers (:1346-1520)
    def _entity_root(self, tile_id: str, round_id: str | None = None, bit_id: str | None = None) -> Path:
        """``<fiducial root>/<tile>/<round>`` or ``<readouts root>/<tile>/<bit>`` (:1346-1376)."""
        if (round_id is None) == (bit_id is None):
            raise ValueError("Provide either round_id or bit_id, but not both.")
        if round_id is not None:
            return self._datastore._fiducial_root_path / Path(tile_id) / Path(round_id)
        return self._datastore._readouts_root_path / Path(tile_id) / Path(bit_id)

    def _shape_of(self, tile_id: str, kind: str, round_id: str | None = None, bit_id: str | None = None):
        """Shape of a stored array or None.  The reference's store answers by path
        (``_image_shape(entity_root / kind)``, :1398-1402); an in-memory store by keys."""
        if (round_id is None) == (bit_id is None):
            raise ValueError("Provide either round_id or bit_id, but not both.")
        ds = self._datastore
        if hasattr(ds, "_fiducial_root_path"):
            return ds._image_shape(self._entity_root(tile_id, round_id=round_id, bit_id=bit_id) / Path(kind))
        return ds._image_shape(tile=tile_id, round=round_id, bit=bit_id, kind=kind)

    def _has_valid_deconvolved_fiducial_image(self, tile_id: str | None = None, round_id: str | None = None) -> bool:
        if round_id is None:
            raise ValueError("round_id is required for fiducial images.")
        tile_id = self._tile_id if tile_id is None else tile_id
        corrected_shape = self._shape_of(tile_id, "corrected_data", round_id=round_id)
        return corrected_shape is not None and self._shape_of(tile_id, "decon_data", round_id=round_id) == corrected_shape

    def _has_valid_deconvolved_readout_image(self, tile_id: str | None = None, bit_id: str | None = None) -> bool:
        tile_id = self._tile_id if tile_id is None else tile_id
        if bit_id is None:
            raise ValueError("bit_id is required for readout images.")
        corrected_shape = self._shape_of(tile_id, "corrected_data", bit_id=bit_id)
        return corrected_shape is not None and self._shape_of(tile_id, "decon_data", bit_id=bit_id) == corrected_shape

    def _has_valid_feature_predictor_outputs(self, tile_id: str | None = None, bit_id: str | None = None) -> bool:
        tile_id = self._tile_id if tile_id is None else tile_id
        if bit_id is None:
            raise ValueError("bit_id is required for feature predictor outputs.")
        ds = self._datastore
        corrected_shape = self._shape_of(tile_id, "corrected_data", bit_id=bit_id)
        folder = getattr(ds, "feature_predictor_folder_name", "feature_predictor")
        feature_shape = self._shape_of(tile_id, f"{folder}_data", bit_id=bit_id)
        if hasattr(ds, "_feature_predictor_localizations_root_path"):
            spots_ok = (ds._feature_predictor_localizations_root_path / Path(tile_id) / Path(bit_id + ".parquet")).exists()
        else:
            spots_ok = ds.has_feature_predictor_spots(tile=tile_id, bit=bit_id)
        return corrected_shape is not None and feature_shape == corrected_shape and spots_ok

    def _is_tile_complete(self, tile_id: str) -> bool:
        if self._decon_fiducial:
            for round_id in self._round_ids:
                if not self._has_valid_deconvolved_fiducial_image(tile_id=tile_id, round_id=round_id):
                    return False
        for round_id in self._round_ids:
            if self._datastore.load_local_round_transform_zyx_um(tile=tile_id, round=round_id) is None:
                return False
        for bit_id in self._bit_ids:
            if not self._has_valid_feature_predictor_outputs(tile_id=tile_id, bit_id=bit_id):
                return False
        return True

    # ----------------------------------------------------------------- entry points (:1522-1591)
    def register_all_tiles(self) -> None:
        tile_ids = list(self._datastore.tile_ids)
        start_idx = 0
        if not self._overwrite_outputs:
            for idx, tile_id in enumerate(tile_ids):
                if self._is_tile_complete(tile_id):
                    start_idx = idx + 1
                    continue
                start_idx = idx
                break
            else:
                start_idx = len(tile_ids)
            if start_idx >= len(tile_ids):
                if self._verbose >= 1:
                    print(time_stamp(), "All tiles already have complete preprocessing outputs.")
                return
            if start_idx > 0 and self._verbose >= 1:
                print(time_stamp(), f"Resuming local registration at tile id: {tile_ids[start_idx]}.")
        todo = tile_ids[start_idx:]
        # One process per GPU: whole tiles are dealt to the ranks first (contiguous ranges, no cross-rank
        # dependency: a round registers against its own tile's round 1); only when there are fewer tiles
        # than ranks are the rounds / bits of a tile shared out instead (the reference's only mode, :2137).
        mine, self._tile_parallel = self._tiles_for_this_rank(todo)
        try:
            for tile_id in mine:
                self.tile_id = tile_id
                self._generate_registrations()
                self._audit_round_transforms()
                self._apply_registration_to_bits()
                # a write-behind store (datastore.ZarrImageDataStore(write_behind=True)) keeps what it was handed
                # readable from HBM while its writer thread works: a finished tile's copies are dropped here - the
                # queued writes hold their own references - or the volumes of a whole dataset would pile up
                evict = getattr(self._datastore, "evict_tile", None)
                if callable(evict):
                    evict(tile_id)
            flush = getattr(self._datastore, "flush", None)
            if callable(flush):
                flush()                       # everything this call produced is on disk when it returns
        finally:
            tile_parallel, self._tile_parallel = self._tile_parallel, False
        count = sum(len(v) for v in self.suspect_registrations.values())
        if tile_parallel:
            import torch
            import torch.distributed as dist

            total = torch.tensor([count], dtype=torch.int64,
                                 device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(total, op=dist.ReduceOp.SUM)      # 8 bytes over RCCL / xGMI: every rank knows the job's count
            count = int(total.item())
            dist.barrier()
        self.suspect_registrations_all_ranks = count
        if count:
            print(time_stamp(), f"WARNING: {count} round registration(s) stand apart from their tile's other rounds by more "
                                "than a quarter of the image (see DataRegistration.suspect_registrations): the phase-correlation "
                                "disambiguation can pick a wrapped twin of the true shift.", flush=True)
        if self._global_registration:
            self.global_register()

    @staticmethod
    def _tiles_for_this_rank(tile_ids: list):
        """(this rank's tiles, tile_parallel).  Under torch.distributed with at least one tile per rank:
        contiguous ceil-sized ranges like PixelDecoder's tile sharding (PixelDecoder.py:4811-4820)."""
        if not _dist_ready():
            return list(tile_ids), False
        import torch.distributed as dist

        world, rank = dist.get_world_size(), dist.get_rank()
        if world <= 1 or len(tile_ids) < world:
            return list(tile_ids), False
        chunk = (len(tile_ids) + world - 1) // world
        return list(tile_ids[rank * chunk: min((rank + 1) * chunk, len(tile_ids))]), True

    def register_one_tile(self, tile_id) -> None:
        self.tile_id = tile_id
        self._generate_registrations()
        self._apply_registration_to_bits()

    def apply_registration_to_one_tile(self, tile_id) -> None:
        self.tile_id = tile_id
        self._apply_registration_to_bits()

    def global_register(self, *args, **kwargs) -> None:
        raise NotImplementedError("global (multiview-stitcher) registration is outside the hot path; "
                                  "run the reference's DataRegistration.global_register on the saved outputs")

    # ----------------------------------------------------------------- work distribution
    def _shares(self, items: list):
        """Contiguous chunks per worker like the reference (:2137-2141); under torch.distributed
        this rank's chunk only."""
        if not items:
            return [], False
        if getattr(self, "_tile_parallel", False):  # this rank owns the whole tile
            return [list(items)], False
        if (not _dist_ready() and min(int(self._num_gpus), len(items)) > 1
                and getattr(self._datastore, "in_memory", False)):
            raise RuntimeError("num_gpus > 1 spawns one child process per GPU, whose writes to an in-memory "
                               "datastore would be lost: use a path-backed datastore, or run one process per GPU "
                               "under torch.distributed")
        if _dist_ready():
            import torch.distributed as dist

            world, rank = dist.get_world_size(), dist.get_rank()
            chunk = (len(items) + world - 1) // world
            return [items[rank * chunk: min((rank + 1) * chunk, len(items))]], True
        workers = max(1, min(int(self._num_gpus), len(items)))
        chunk = (len(items) + workers - 1) // workers
        return [items[g * chunk: min((g + 1) * chunk, len(items))] for g in range(workers)], False

    def _run_round_workers(self, target, items: list, extra_args: tuple, label: str) -> list:
        """Run ``target(self, subset, *extra, gpu, queue)`` over the shares; returns payloads."""
        import torch

        shares, distributed = self._shares(items)
        if distributed or len(shares) <= 1:
            q = _LocalQueue()
            if shares and shares[0]:
                target(self, shares[0], *extra_args, torch.cuda.current_device(), q)
            messages = q.items
            if distributed:
                import torch.distributed as dist

                dist.barrier()
            expected = shares[0] if shares else []
        else:
            import torch.multiprocessing as mp

            from .PixelDecoder import _start_gpu_worker_process

            ctx = mp.get_context("spawn")
            result_queue = ctx.Queue()
            processes = [
                _start_gpu_worker_process(target=target, args=(self, subset, *extra_args, 0, result_queue),
                                          physical_gpu_id=gpu)
                for gpu, subset in enumerate(shares) if subset
            ]
            messages, expected = [], items
            while len(messages) < len(items):
                try:
                    messages.append(result_queue.get(timeout=5))
                except queue.Empty:
                    if any(p.exitcode not in (None, 0) for p in processes) or all(
                            p.exitcode is not None for p in processes):
                        break
            errors = [m[2] for m in messages if m[0] != "result"]
            for p in processes:
                p.join()
                if p.exitcode not in (0, None):
                    errors.append(f"{label} worker pid={p.pid} failed with exitcode={p.exitcode}.")
            if errors:
                raise RuntimeError(f"{label} failed:\n" + "\n".join(str(e) for e in errors))
        done = {m[1] for m in messages if m[0] == "result"}
        missing = [r for r in expected if r not in done]
        errors = [m[2] for m in messages if m[0] != "result"]
        if missing:
            errors.append(f"Missing processed rounds: {missing}")
        if errors:
            raise RuntimeError(f"{label} failed:\n" + "\n".join(str(e) for e in errors))
        return [m[2] for m in messages if m[0] == "result"]

    # ----------------------------------------------------------------- per-tile stages
    def _generate_registrations(self) -> None:
        """Deconvolve fiducials, register rounds, save transforms and flow fields (:2086-2270)."""
        import torch

        if self._num_gpus == 0:
            raise RuntimeError("No GPUs detected. Cannot run _generate_registrations().")
        start_time = timeit.default_timer()
        reference_round_id = self._round_ids[0]
        gpu_id = torch.cuda.current_device()
        reference_image = _load_deconvolve_fiducial_round(dr=self, round_id=reference_round_id, gpu_id=gpu_id)
        self._datastore.save_local_round_transform_zyx_um(np.eye(4, dtype=np.float32), tile=self._tile_id,
                                                          round=reference_round_id)
        if self._verbose >= 1:
            print(time_stamp(), f"Finished fiducial tile id: {self._tile_id}; round id: {reference_round_id}.")
        moving_rounds = list(self._round_ids[1:])
        if not moving_rounds:
            del reference_image
            _release_worker_gpu_memory()
            return
        spawn = not _dist_ready() and min(int(self._num_gpus), len(moving_rounds)) > 1
        payloads = self._run_round_workers(
            _process_fiducial_rounds_on_gpu, moving_rounds,
            (reference_image.cpu().numpy() if spawn else reference_image,), "Fiducial registration")
        requested_crop_yx = min(int(self._crop_yx_decon), int(max(reference_image.shape[-2:])))
        for payload in payloads:
            if payload is not None and int(payload) < requested_crop_yx:
                self._crop_yx_decon = int(payload)
        del reference_image
        _release_worker_gpu_memory()
        if self._perform_deformable_registration:
            self._run_round_workers(_process_sofima_rounds_on_gpu, moving_rounds, (), "SOFIMA registration")
        _registration_diag(f"fiducial_registration_all_done tile={self._tile_id} rounds={len(self._round_ids)} "
                           f"elapsed_s={timeit.default_timer() - start_time:.2f}",
                           enabled=self._registration_diagnostics)
        gc.collect()

    def _audit_round_transforms(self) -> list:
        """Make mis-registrations visible (parity behaviour unchanged: nothing is corrected).

        ``phase_cross_correlation(disambiguate=True)`` - scikit-image's rule, restated here as the reference calls it
        (multiview_registration.py:265-301) - chooses between a shift and its wrapped twins by correlating the
        overlap tiles and accepts tiles of three samples; on the 2-D projection stage a shift of about (+-3, +-1) px
        gives such a corner tile, which wins about one time in seven and registers the round an image width off
        (on the F1 fixture of tests/golden/f1_reference_run.json it happens to round 6).  A
        wrapped twin differs from the truth by about the image extent, real round-to-round drifts are a few pixels:
        a round whose translation is further than a QUARTER of the image extent (any axis) from the median over the
        tile's rounds is recorded in ``self.suspect_registrations[tile_id]`` as ``(round_id, shift_px, median_px)``.
        ``register_all_tiles`` sums the count over all ranks and prints one warning."""
        ds = self._datastore
        spacing = np.asarray(ds.voxel_size_zyx_um, dtype=np.float64)
        shifts, rounds = [], []
        for round_id in self._round_ids:
            T = ds.load_local_round_transform_zyx_um(tile=self._tile_id, round=round_id)
            if T is None:
                continue
            shifts.append(np.asarray(T, dtype=np.float64)[:3, 3] / spacing)
            rounds.append(round_id)
        suspects = []
        if len(shifts) >= 3:
            shape = self._shape_of(self._tile_id, "corrected_data", round_id=self._round_ids[0])
            if shape is not None:
                extent = np.asarray(tuple(shape)[-3:], dtype=np.float64)
                shifts = np.asarray(shifts)
                median = np.median(shifts, axis=0)
                far = np.any(np.abs(shifts - median) > 0.25 * extent, axis=1)
                suspects = [(rounds[i], shifts[i].round(2).tolist(), median.round(2).tolist()) for i in np.nonzero(far)[0]]
        if suspects:
            self.suspect_registrations[self._tile_id] = suspects
        else:
            self.suspect_registrations.pop(self._tile_id, None)
        return suspects

    def _apply_registration_to_bits(self) -> None:
        """Deconvolve readout bits, run the feature predictor, save unwarped outputs (:2272-2315)."""
        import torch

        if self._num_gpus == 0:
            raise RuntimeError("No GPUs detected. Cannot run _apply_registration_to_bits().")
        shares, distributed = self._shares(list(self._bit_ids))
        if distributed or len(shares) <= 1:
            if shares and shares[0]:
                _apply_bits_on_gpu(self, shares[0], torch.cuda.current_device())
            if distributed:
                import torch.distributed as dist

                dist.barrier()
            return
        from .PixelDecoder import _join_gpu_workers, _start_gpu_worker_process

        processes = [_start_gpu_worker_process(target=_apply_bits_on_gpu, args=(self, subset, 0), physical_gpu_id=gpu)
                     for gpu, subset in enumerate(shares) if subset]
        _join_gpu_workers(processes, "Readout preprocessing")
