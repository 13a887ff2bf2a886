"""PixelDecoder on MI355X - per-tile MERFISH pixel decoding (mirror of PixelDecoder.py).

Same constructor, method names, defaults and result types as the reference operator
(``PixelDecoder.py:417-557, 4471-4870``); the datastore is used through the same duck-type
(SURVEY.md 8b B3).  What differs is where the data lives: one tile's working set (the
``[B,Z,Y,X]`` float32 bit stack, the float16 / int16 result volumes) stays in HBM from load
to transcript table; the reference bounces every stage through host NumPy.  Device work:

* ``_load_bit_data``   ``m3d_weight_readout`` (+ ``m3d_affine_warp`` / ``m3d_flow_warp``)
* ``_lp_filter``       ``m3d_gaussian_lowpass``
* ``_decode_pixels``   ``m3d_decode_pixels`` - scale, clip, L2 normalise, exact-f32 MFMA
                       distance GEMM, argmin, two-threshold call in one pass
* ``_extract_barcodes`` ``m3d_ccl_label`` + ``m3d_region_stats``; only the compact
                       per-component table crosses PCIe, derived columns are NumPy on host
* ``_global_normalization_vectors``  ``m3d_radix_hist`` exact order statistics, summed over
                       ranks with one RCCL all-reduce when tiles are sharded across GPUs

The host-side tail of ``decode_all_tiles`` / ``optimize_filtering`` (blank-fraction filter, duplicate
removal, LR filter) lives in ``postdecode.py``.  Out of scope here:
cell assignment from ROIs, chromatic-affine estimation, napari display.
"""

from __future__ import annotations

import gc
import hashlib
import operator
import os
import shutil
import tempfile
import warnings
from collections.abc import Sequence
from pathlib import Path
from random import sample
from typing import Literal

import numpy as np
import pandas as pd

from . import kernels
from .utils.decode_warping import warp_bit_image_to_reference

DEFAULT_DECODE_LOWPASS_SIGMA = (3.0, 1.0, 1.0)
DEFAULT_DECODE_MAGNITUDE_THRESHOLD = (1.5, 10.0)
DEFAULT_2D_MINIMUM_PIXELS = 7.0
DEFAULT_3D_MINIMUM_PIXELS = 16.0


def time_stamp() -> str:
    import datetime

    return datetime.datetime.now().strftime("%Y_%m_%d_%H:%M:%S")


def _torch():
    import torch

    return torch


def _device(gpu_id: int = 0):
    return _torch().device("cuda", int(gpu_id))


def _to_numpy(x):
    torch = _torch()
    return x.cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def _to_tensor(x, device, dtype=None):
    torch = _torch()
    if isinstance(x, torch.Tensor):
        t = x.to(device)
    else:
        t = torch.from_numpy(np.ascontiguousarray(x)).to(device)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _result(value):
    """Resolve the tensorstore-style future the datastore returns by default (:1873-1874)."""
    return value.result() if hasattr(value, "result") else value


# ----------------------------------------------------------------------------------------
# multi-GPU helpers (PixelDecoder.py:142-204): one child per GPU, device pinned by env
# ----------------------------------------------------------------------------------------
def _start_gpu_worker_process(*, target, args: tuple, physical_gpu_id: int):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    previous = os.environ.get("HIP_VISIBLE_DEVICES")
    os.environ["HIP_VISIBLE_DEVICES"] = str(int(physical_gpu_id))
    try:
        process = ctx.Process(target=target, args=args)
        process.start()
    finally:
        if previous is None:
            os.environ.pop("HIP_VISIBLE_DEVICES", None)
        else:
            os.environ["HIP_VISIBLE_DEVICES"] = previous
    return process


def _join_gpu_workers(processes, label: str) -> None:
    errors = []
    for process in processes:
        process.join()
        if process.exitcode not in (0, None):
            errors.append(f"{label} worker pid={process.pid} failed with exitcode={process.exitcode}.")
    if errors:
        raise RuntimeError(f"{label} failed:\n" + "\n".join(errors))


def _open_datastore(datastore_path):
    """Re-open the on-disk store inside a worker (PixelDecoder.py:262-270): this package's reader of the
    reference's directory layout, images loaded straight into the worker's GPU."""
    from .qi2lab_store import Qi2labDiskStore

    torch = _torch()
    device = f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else None
    return Qi2labDiskStore(datastore_path, device=device)


def decode_tiles_worker(datastore_path, tile_indices, gpu_id, merfish_bits, verbose, decode_mode,
                        lowpass_sigma, magnitude_threshold, minimum_pixels,
                        feature_predictor_threshold, normalization_method) -> None:
    """decode_one_tile over a tile subset on one GPU (PixelDecoder.py:207-309)."""
    _torch().cuda.set_device(gpu_id)
    decoder = PixelDecoder(datastore=_open_datastore(datastore_path), use_mask=False,
                           merfish_bits=merfish_bits, num_gpus=1, verbose=0, decode_mode=decode_mode)
    decoder._load_global_normalization_vectors(gpu_id=gpu_id, lowpass_sigma=lowpass_sigma)
    decoder._load_iterative_normalization_vectors(gpu_id=gpu_id)
    decoder._optimize_normalization_weights = False
    for k, tile_idx in enumerate(tile_indices):
        if verbose >= 1:
            print(time_stamp(), f"GPU {gpu_id}: starting tile {k + 1} of {len(tile_indices)} "
                  f"(tile index: {tile_idx}).", flush=True)
        decoder.decode_one_tile(tile_idx=tile_idx, display_results=False, return_results=False,
                                lowpass_sigma=lowpass_sigma, magnitude_threshold=magnitude_threshold,
                                minimum_pixels=minimum_pixels,
                                feature_predictor_threshold=feature_predictor_threshold,
                                normalization_method=normalization_method, gpu_id=gpu_id)
        decoder._save_barcodes()
        decoder._cleanup()


def _optimize_norm_worker(datastore_path, tile_indices, gpu_id, merfish_bits, decode_mode, temp_dir,
                          iteration, lowpass_sigma, magnitude_threshold, minimum_pixels,
                          feature_predictor_threshold, collect_chromatic_centroids,
                          excluded_gene_ids) -> None:
    """One normalisation-by-decoding iteration on one GPU (PixelDecoder.py:312-414)."""
    _torch().cuda.set_device(gpu_id)
    decoder = PixelDecoder(datastore=_open_datastore(datastore_path), use_mask=False,
                           merfish_bits=merfish_bits, num_gpus=1, verbose=0, decode_mode=decode_mode,
                           excluded_gene_ids=excluded_gene_ids)
    decoder._run_optimize_iteration(tile_indices, gpu_id, temp_dir, iteration, lowpass_sigma,
                                    magnitude_threshold, minimum_pixels, feature_predictor_threshold)


class PixelDecoder:
    """Retrieve and process one tile at a time from a qi2lab-style datastore."""

    def __init__(
        self,
        datastore,
        merfish_bits: int = 16,
        num_gpus: int = 1,
        verbose: int = 1,
        use_mask: bool | None = False,
        z_range: Sequence[int] | None = None,
        decode_mode: Literal["auto", "2d", "3d"] = "auto",
        estimate_chromatic_affines: bool = False,
        chromatic_affine_config=None,
        excluded_gene_ids: Sequence[str] | None = None,
    ) -> None:
        self._datastore_path = Path(datastore._datastore_path)
        self._datastore = datastore
        self._num_gpus = num_gpus
        self._verbose = verbose
        self._barcodes_filtered = False
        self._n_merfish_bits = merfish_bits
        if decode_mode not in {"auto", "2d", "3d"}:
            raise ValueError("decode_mode must be one of 'auto', '2d', or '3d'.")
        self._decode_mode = decode_mode
        if estimate_chromatic_affines:
            raise NotImplementedError(
                "chromatic-affine estimation from barcodes is outside the hot path "
                "(reference default is False, PixelDecoder.py:458)")
        self._estimate_chromatic_affines = False
        self._chromatic_affine_config = chromatic_affine_config
        if decode_mode == "auto":
            effective = "2d" if self._datastore.microscope_type == "2D" else "3d"
        else:
            effective = decode_mode
        self._effective_decode_mode = effective
        self._is_3D = effective != "2d"
        self._decode_run_key = None
        if z_range is None:
            self._z_crop = False
            self._z_range = [0, None]
        else:
            self._z_crop = True
            self._z_range = [z_range[0], z_range[1]]
        self._z_slice = slice(self._z_range[0], self._z_range[1])

        self._load_codebook()
        self._excluded_gene_ids, self._excluded_codeword_indices = self._resolve_excluded_gene_ids(
            excluded_gene_ids)
        self._optimization_excluded_gene_ids = self._excluded_gene_ids
        self._decoding_matrix_no_errors = self._normalize_codebook(include_errors=False)
        self._decoding_matrix = self._decoding_matrix_no_errors.copy()
        self._barcode_count = self._decoding_matrix.shape[0]
        self._bit_count = self._decoding_matrix.shape[1]
        if use_mask:
            raise NotImplementedError("use_mask is not implemented in the reference either (:543)")
        self._mask_image = None
        self._codebook_style = 1
        self._optimize_normalization_weights = False
        self._collect_chromatic_centroids = False
        self._global_normalization_loaded = False
        self._iterative_normalization_loaded = False
        self._global_normalization_vector = None
        self._global_background_vector = None
        self._iterative_normalization_vector = None
        self._iterative_background_vector = None
        self._load_tile_decoding = False
        self._filter_type = "raw"
        self._df_barcodes = pd.DataFrame()
        self._results = {}

    # ------------------------------------------------------------------ codebook (:756-932)
    def _load_codebook(self) -> None:
        """Codebook, thresholds from the median on-bit count, blanks (:756-800)."""
        self._df_codebook = self._datastore.codebook.copy()
        self._df_codebook.fillna(0, inplace=True)
        bit_columns = self._df_codebook.columns[1 : self._n_merfish_bits + 1]
        on_counts = self._df_codebook.loc[:, bit_columns].to_numpy(dtype=np.int8, copy=False).sum(axis=1)
        self._df_codebook = self._df_codebook.loc[on_counts != 1].reset_index(drop=True)
        self._codebook_matrix = self._df_codebook.loc[:, bit_columns].to_numpy(dtype=int, copy=False)
        k = int(np.median(on_counts[on_counts != 1]))
        self._pixel_assignment_threshold = float(np.sqrt(2.0 - 2.0 * ((k - 2.0) / np.sqrt(k * (k - 2.0)))))
        self._transcript_distance_threshold = float(np.sqrt(2.0 - 2.0 * (k / np.sqrt(k * (k + 2.0)))))
        self._blank_count = int(self._df_codebook["gene_id"].astype("string").str.lower()
                                .str.startswith("blank", na=False).sum())
        self._gene_ids = self._df_codebook.iloc[:, 0].tolist()

    def _resolve_excluded_gene_ids(self, excluded_gene_ids):
        """Requested gene IDs -> stable full-codebook row indices (:802-841)."""
        if not excluded_gene_ids:
            return (), ()
        requested, seen = [], set()
        for value in excluded_gene_ids:
            gene_id = str(value).strip()
            if not gene_id or gene_id in seen:
                continue
            requested.append(gene_id)
            seen.add(gene_id)
        index_by_gene: dict[str, list[int]] = {}
        for index, value in enumerate(self._gene_ids):
            index_by_gene.setdefault(str(value), []).append(index)
        unknown = [g for g in requested if g not in index_by_gene]
        if unknown:
            raise ValueError("Optimization exclusion gene IDs are not present in the active "
                             f"codebook: {', '.join(unknown)}. Matching is case-sensitive.")
        excluded_indices = tuple(i for g in requested for i in index_by_gene[g])
        if len(excluded_indices) >= len(self._gene_ids):
            raise ValueError("Optimization exclusions cannot remove every codeword.")
        wanted = set(requested)
        resolved = tuple(dict.fromkeys(str(g) for g in self._gene_ids if str(g) in wanted))
        return resolved, excluded_indices

    def _codebook_fingerprint(self) -> str:
        """SHA-256 of active codebook IDs and bits (:843-853)."""
        digest = hashlib.sha256()
        matrix = np.ascontiguousarray(self._codebook_matrix, dtype=np.int8)
        digest.update(np.asarray(matrix.shape, dtype=np.int64).tobytes())
        for gene_id in self._gene_ids:
            encoded = str(gene_id).encode("utf-8")
            digest.update(len(encoded).to_bytes(8, byteorder="little"))
            digest.update(encoded)
        digest.update(matrix.tobytes())
        return digest.hexdigest()

    def _iterative_normalization_metadata(self) -> dict[str, object]:
        return {"scope": "iterative_optimization",
                "excluded_gene_ids": list(self._optimization_excluded_gene_ids),
                "codebook_sha256": self._codebook_fingerprint()}

    @staticmethod
    def _suppress_excluded_codeword_assignments(decoded_trace, codebook_index_trace,
                                                excluded_codeword_indices) -> None:
        """Excluded winners become background, no fall-through to the runner-up (:863-877)."""
        if not excluded_codeword_indices:
            return
        torch = _torch()
        if isinstance(decoded_trace, torch.Tensor):
            ex = torch.as_tensor(list(excluded_codeword_indices), dtype=codebook_index_trace.dtype,
                                 device=codebook_index_trace.device)
            decoded_trace[torch.isin(codebook_index_trace, ex)] = -1
        else:
            ex = np.asarray(excluded_codeword_indices, dtype=codebook_index_trace.dtype)
            decoded_trace[np.isin(codebook_index_trace, ex)] = -1

    def _normalize_codebook(self, gpu_id: int = 0, include_errors: bool = False) -> np.ndarray:
        """Unit-L2 codewords, float64 on the host (:879-932; C x B values - not device work)."""
        barcode_set = np.asarray(self._codebook_matrix[:, 0 : self._n_merfish_bits])
        magnitudes = np.linalg.norm(barcode_set, axis=1, keepdims=True)
        magnitudes[magnitudes == 0] = 1
        normalized = barcode_set / magnitudes
        if not include_errors:
            return normalized
        out = [normalized]
        for bit_index in range(barcode_set.shape[1]):
            flipped = barcode_set.copy()
            flipped[:, bit_index] = 1 - flipped[:, bit_index]
            mags = np.sqrt(np.sum(flipped**2, axis=1))
            mags = np.where(mags == 0, 1, mags)
            out.append(flipped / mags[:, None])
        return np.vstack(out)

    # ------------------------------------------------------------ normalisation (:725-1421)
    @staticmethod
    def _normalization_pixels(image, cell_mask):
        """All voxels, or only voxels whose YX centre lies in a cell (:725-730)."""
        if cell_mask is None:
            return image.reshape(-1)
        return image[:, cell_mask].reshape(-1)

    def _load_global_normalization_vectors(self, gpu_id: int = 0, recalculate: bool = False,
                                           tile_indices=None,
                                           lowpass_sigma=DEFAULT_DECODE_LOWPASS_SIGMA) -> None:
        """Load cached global vectors or compute them (:934-979)."""
        norm, bkd = self._datastore.load_decode_normalization_vectors(self._decode_run_key, "global")
        if not recalculate and norm is not None and bkd is not None:
            self._global_normalization_vector = np.asarray(norm, dtype=np.float32)
            self._global_background_vector = np.asarray(bkd, dtype=np.float32)
            self._global_normalization_loaded = True
        else:
            self._global_normalization_vectors(gpu_id=gpu_id, tile_indices=tile_indices,
                                               lowpass_sigma=lowpass_sigma)

    def _weighted_warped_bit(self, tile_id, bit_id, gpu_id: int):
        """float32(readout) * float32(predictor), warped to the reference frame (:1047-1070)."""
        torch = _torch()
        dev = _device(gpu_id)
        readout = _result(self._datastore.load_local_readout_image(tile=tile_id, bit=bit_id,
                                                                  return_future=False))
        predictor = _result(self._datastore.load_local_feature_predictor_image(
            tile=tile_id, bit=bit_id, return_future=False))
        _ex, em = self._datastore.load_local_wavelengths_um(tile=tile_id, bit=bit_id)
        r = _to_tensor(readout, dev)
        p = _to_tensor(predictor, dev, torch.float32)
        if r.dtype == torch.uint16:
            weighted = kernels.weight_readout(r, p)
        else:  # already float (e.g. a float32 deconvolved image): same product in float32
            weighted = r.to(torch.float32) * p
        warped = warp_bit_image_to_reference(weighted, datastore=self._datastore, tile=tile_id,
                                             bit_id=bit_id, emission_wavelength_um=em, gpu_id=gpu_id)
        return warped, em

    def _global_normalization_vectors(self, low_percentile_cut: float = 10.0,
                                      high_percentile_cut: float = 90.0,
                                      hot_pixel_threshold: int = 50000, gpu_id: int = 0,
                                      tile_indices=None,
                                      lowpass_sigma=DEFAULT_DECODE_LOWPASS_SIGMA) -> None:
        """Per-bit background / foreground vectors from percentile tails (:981-1199).

        Per tile: weighted + warped image, hot pixels (> 50000) -> median of the middle z
        plane, z-slice, low-pass.  background = median of the values strictly below each
        tile's 10th percentile, pooled; normalisation = median of ``max(x - bkd, 0)``
        strictly above each tile's 90th percentile, pooled.  With ``torch.distributed``
        initialised the tiles are sharded over ranks and the pooled medians are exact
        distributed radix selects (one 2-KB RCCL all-reduce per pass).
        """
        torch = _torch()
        sigma = self._effective_lowpass_sigma(lowpass_sigma)
        if tile_indices is not None:
            tiles = [self._datastore.tile_ids[i] for i in tile_indices]
        elif len(self._datastore.tile_ids) > 5:
            tiles = sample(self._datastore.tile_ids, 5)  # unseeded, like the reference (:1018)
        else:
            tiles = list(self._datastore.tile_ids)
        group = False
        if kernels._dist_ready():
            import torch.distributed as dist

            tiles = tiles[dist.get_rank()::dist.get_world_size()]
            group = None
        n_all = len(self._datastore.bit_ids)
        normalization_vector = np.ones(n_all, dtype=np.float32)
        background_vector = np.zeros(n_all, dtype=np.float32)
        dev = _device(gpu_id)
        for bit_idx, bit_id in enumerate(self._datastore.bit_ids):
            images = []
            for tile_id in tiles:
                img, _em = self._weighted_warped_bit(tile_id, bit_id, gpu_id)
                img = img.contiguous()
                mid = kernels.PooledValues(group=False)
                mid.add(img[img.shape[0] // 2].contiguous())
                kernels.replace_above(img, float(hot_pixel_threshold), mid.median())
                img = img[self._z_slice].contiguous()
                images.append(self._lowpass_image(img, sigma=sigma))
            low = kernels.PooledValues(group=group)
            for img in images:
                per_tile = kernels.PooledValues(group=False)
                per_tile.add(img)
                low.add(img, below=per_tile.percentile(low_percentile_cut))
            if not images:
                low.add(torch.empty(0, dtype=torch.float32, device=dev))
            background_vector[bit_idx] = low.median() if low.count() > 0 else 0
            bkd = float(background_vector[bit_idx])
            high = kernels.PooledValues(group=group)
            for img in images:
                per_tile = kernels.PooledValues(group=False)
                per_tile.add(img, sub=bkd)
                high.add(img, sub=bkd, above=per_tile.percentile(high_percentile_cut))
            if not images:
                high.add(torch.empty(0, dtype=torch.float32, device=dev))
            normalization_vector[bit_idx] = high.median() if high.count() > 0 else 1
            del images
        self._datastore.save_decode_normalization_vectors(
            self._decode_run_key, "global", normalization_vector.astype(np.float32),
            background_vector.astype(np.float32), decode_mode=self._effective_decode_mode)
        self._global_background_vector = background_vector
        self._global_normalization_vector = normalization_vector
        self._global_normalization_loaded = True

    def _load_iterative_normalization_vectors(self, gpu_id: int = 0) -> None:
        """Load cached iterative vectors (codebook-fingerprint guarded) or compute (:1201-1248)."""
        load_metadata = getattr(self._datastore, "load_decode_normalization_metadata", None)
        metadata = load_metadata(self._decode_run_key, "iterative") if callable(load_metadata) else None
        expected = metadata.get("codebook_sha256") if isinstance(metadata, dict) else None
        if expected is not None and expected != self._codebook_fingerprint():
            raise ValueError("Cached iterative normalization vectors were fitted with a "
                             "different active codebook. Re-run iterative optimization.")
        norm, bkd = self._datastore.load_decode_normalization_vectors(self._decode_run_key, "iterative")
        if norm is not None and bkd is not None:
            self._iterative_background_vector = np.nan_to_num(np.asarray(bkd), 0.0)
            self._iterative_normalization_vector = np.nan_to_num(np.asarray(norm), 1.0)
            self._iterative_normalization_loaded = True
        else:
            self._iterative_normalization_vectors(gpu_id=gpu_id)

    def _iterative_normalization_vectors(self, gpu_id: int = 0) -> None:
        """Medians of on-bit / off-bit mean intensities of non-blank barcodes (:1250-1421).

        Host-side table arithmetic in the reference too; the per-row ``iterrows`` loop
        (:1315-1333) is replaced by one masked NumPy median with identical results.
        """
        df = self._df_barcodes_loaded
        keep = ~df["gene_id"].astype("string").str.lower().str.startswith("blank", na=False)
        df = df[keep]
        if self._iterative_background_vector is None and self._iterative_normalization_vector is None:
            old_bkd = np.round(np.asarray(self._global_background_vector[: self._n_merfish_bits]), 1)
            old_norm = np.round(np.asarray(self._global_normalization_vector[: self._n_merfish_bits]), 1)
        else:
            old_bkd = np.asarray(self._iterative_background_vector)
            old_norm = np.asarray(self._iterative_normalization_vector)
        bit_columns = sorted(c for c in df.columns if c.startswith("bit") and c.endswith("_mean_intensity"))
        if df.empty or not bit_columns:
            self._datastore.save_decode_normalization_vectors(
                self._decode_run_key, "iterative", old_norm.astype(np.float32),
                old_bkd.astype(np.float32), decode_mode=self._effective_decode_mode,
                metadata=self._iterative_normalization_metadata())
            return
        values = df[bit_columns].to_numpy(dtype=np.float64)
        col_bits = np.asarray([int(c[3:5]) for c in bit_columns])
        on = df[["on_bit_1", "on_bit_2", "on_bit_3", "on_bit_4"]].to_numpy(dtype=np.int64)
        is_on = (col_bits[None, :, None] == on[:, None, :]).any(axis=2)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)
            sig = np.nanmedian(np.where(is_on, values, np.nan), axis=0)
            bkg = np.nanmedian(np.where(~is_on, values, np.nan), axis=0)
        norm = np.round(sig.astype(np.float32), 1)
        bkd = np.round(bkg.astype(np.float32), 1)
        norm = np.nan_to_num(norm, 1.0)
        norm = np.where(norm == 0.0, 1.0, norm)
        bkd = np.nan_to_num(bkd, 0.0)
        if self._verbose > 1:
            print(time_stamp(), "Normalizations updated.")
            print(f"Background delta: {np.round(np.abs(bkd - old_bkd), 1)}")
            print(f"Foreground delta: {np.round(np.abs(norm - old_norm), 1)}")
            print(f"Num. barcodes: {len(df)}")
        self._iterative_normalization_vector = norm
        self._iterative_background_vector = bkd
        self._datastore.save_decode_normalization_vectors(
            self._decode_run_key, "iterative", norm.astype(np.float32), bkd.astype(np.float32),
            decode_mode=self._effective_decode_mode, metadata=self._iterative_normalization_metadata())
        self._iterative_normalization_loaded = True

    # ---------------------------------------------------------------- data load (:1828-2041)
    def _load_bit_data(self, feature_predictor_threshold: float | None = 0.1, gpu_id: int = 0) -> None:
        """readout x predictor per bit, warped, z-sliced, stacked in HBM (:1828-1946).

        ``feature_predictor_threshold`` is the reference's dead legacy argument (:1837-1839).
        """
        torch = _torch()
        bit_ids = self._datastore.bit_ids[0 : self._n_merfish_bits]
        images, self._em_wvl = [], []
        for bit_id in bit_ids:
            warped, em = self._weighted_warped_bit(self._tile_idx, bit_id, gpu_id)
            images.append(warped[self._z_slice])
            self._em_wvl.append(em)
        self._image_data = torch.stack(images, dim=0).contiguous()
        del images
        if self._decode_mode == "3d" and self._image_data.shape[1] < 2:
            raise ValueError("decode_mode='3d' requires at least two z planes after applying z_range.")
        voxel = self._datastore.voxel_size_zyx_um
        self._pixel_size = voxel[1]
        self._axial_step = voxel[0]
        stage = self._datastore.load_local_stage_position_zyx_um(tile=self._tile_idx, round=0)
        stage_origin = None
        camera_to_stage = np.eye(4, dtype=np.float32)
        if stage is not None:
            stage_origin, camera_to_stage = stage
            stage_origin = np.asarray(stage_origin, dtype=np.float32)
            camera_to_stage = np.asarray(camera_to_stage, dtype=np.float32)
        affine, origin, spacing = self._datastore.load_global_coord_xforms_um(tile=self._tile_idx)
        if affine is None or origin is None or spacing is None:
            affine = np.eye(4)
            if stage_origin is None:
                origin = np.zeros(3, dtype=np.float32)
            elif not self._is_3D and stage_origin.size == 2:
                origin = np.asarray([0, stage_origin[0], stage_origin[1]], dtype=np.float32)
            else:
                origin = stage_origin
            spacing = self._datastore.voxel_size_zyx_um
        self._affine = np.asarray(affine, dtype=np.float32)
        self._origin = np.asarray(origin, dtype=np.float32)
        self._spacing = np.asarray(spacing, dtype=np.float32)
        self._camera_to_stage_affine = camera_to_stage

    def _lowpass_image(self, image, sigma):
        """Gaussian low-pass of one bit image; skipped when sigma is None or has a zero (:1948-1980)."""
        if sigma is None or np.any(np.asarray(sigma, dtype=float) == 0):
            return image
        return kernels.gaussian_lowpass(image, sigma, is_3d=self._is_3D)

    def _lp_filter(self, gpu_id: int = 0, sigma=DEFAULT_DECODE_LOWPASS_SIGMA) -> None:
        """Low-pass every bit of the tile in HBM (:1982-2024)."""
        torch = _torch()
        self._image_data_lp = torch.empty_like(self._image_data)
        scratch = torch.empty_like(self._image_data[0])
        for b in range(self._image_data.shape[0]):
            if sigma is None or np.any(np.asarray(sigma, dtype=float) == 0):
                self._image_data_lp[b].copy_(self._image_data[b])
            else:
                kernels.gaussian_lowpass(self._image_data[b], sigma, is_3d=self._is_3D,
                                         out=self._image_data_lp[b], scratch=scratch)
        self._filter_type = "lp"
        del self._image_data

    def _effective_lowpass_sigma(self, sigma):
        if sigma is None:
            return None
        sigma_zyx = tuple(float(v) for v in sigma)
        if len(sigma_zyx) != 3:
            raise ValueError("lowpass_sigma must contain three values: z, y, x.")
        return sigma_zyx

    def _default_minimum_pixels(self) -> float:
        return DEFAULT_3D_MINIMUM_PIXELS if self._is_3D else DEFAULT_2D_MINIMUM_PIXELS

    # -------------------------------------------------------------- pixel decode (:2354-2643)
    @staticmethod
    def _scale_pixel_traces(pixel_traces, background_vector, normalization_vector,
                            merfish_bits: int = 16, gpu_id: int = 0):
        """(x - bkd) / norm per bit (:2354-2401).  Stand-alone helper; ``_decode_pixels`` fuses it."""
        torch = _torch()
        dev = _device(gpu_id)
        x = _to_tensor(pixel_traces, dev, torch.float32)
        b = _to_tensor(background_vector, dev, torch.float32)[0:merfish_bits]
        n = _to_tensor(normalization_vector, dev, torch.float32)[0:merfish_bits]
        return (x - b[:, None]) / n[:, None]

    @staticmethod
    def _clip_pixel_traces(pixel_traces, clip_lower: float = 0.0, clip_upper: float = 1.0, gpu_id: int = 0):
        """In-place clip (:2403-2433)."""
        return pixel_traces.clamp_(clip_lower, clip_upper)

    @staticmethod
    def _normalize_pixel_traces(pixel_traces, gpu_id: int = 0):
        """L2 normalise over bits; zero norm -> trace 0, reported norm -1 (:2435-2468)."""
        torch = _torch()
        x = _to_tensor(pixel_traces, _device(gpu_id), torch.float32)
        norms = torch.linalg.norm(x, dim=0)
        norms = torch.where(norms == 0, torch.full_like(norms, float("inf")), norms)
        normalized = x / norms
        norms = torch.where(torch.isinf(norms), torch.full_like(norms, -1.0), norms)
        return normalized, norms

    @staticmethod
    def _calculate_distances(pixel_traces, codebook_matrix, gpu_id: int = 0):
        """Min distance / argmin to the codebook for already-normalised traces (:2470-2521).

        Runs the same exact-f32 MFMA kernel as ``_decode_pixels`` (no threshold applied).
        """
        torch = _torch()
        dev = _device(gpu_id)
        x = _to_tensor(pixel_traces, dev, torch.float32)
        cb = _to_tensor(np.asarray(codebook_matrix, dtype=np.float32)
                        if not isinstance(codebook_matrix, torch.Tensor) else codebook_matrix,
                        dev, torch.float32)
        stack = x.reshape(x.shape[0], 1, 1, -1).contiguous()
        dec, _m, _d, _s, _m32, d32 = kernels.decode_pixels(
            stack, cb, float("inf"), (-float("inf"), float("inf")), want_f32=True, prenormalized=True)
        return d32.reshape(-1), dec.reshape(-1).to(torch.int64)

    def _decode_pixels(self, magnitude_threshold: Sequence[float] = (1.1, 2.0), gpu_id: int = 0) -> None:
        """Fused per-pixel decode of the whole tile (:2523-2643)."""
        torch = _torch()
        stack = self._image_data_lp if self._filter_type == "lp" else self._image_data
        bkd = nrm = None
        if self._iterative_normalization_loaded:
            bkd, nrm = self._iterative_background_vector, self._iterative_normalization_vector
        elif self._global_normalization_loaded:
            bkd, nrm = self._global_background_vector, self._global_normalization_vector
        dev = stack.device
        kw = {}
        if bkd is not None:
            kw = dict(background=_to_tensor(np.asarray(bkd, dtype=np.float32), dev),
                      normalization=_to_tensor(np.asarray(nrm, dtype=np.float32), dev))
        decoded, magnitude, distance, scaled = kernels.decode_pixels(
            stack, torch.from_numpy(np.asarray(self._decoding_matrix, dtype=np.float32)),
            self._pixel_assignment_threshold, magnitude_threshold,
            excluded_indices=tuple(self._excluded_codeword_indices), **kw)
        self._results = {"decoded": decoded, "magnitude": magnitude, "distance": distance,
                         "scaled": scaled}

    # result volumes: HBM tensors inside, NumPy through the reference's attribute names
    @property
    def _decoded_image(self):
        return _to_numpy(self._results["decoded"])

    @property
    def _magnitude_image(self):
        return _to_numpy(self._results["magnitude"])

    @property
    def _distance_image(self):
        return _to_numpy(self._results["distance"])

    @property
    def _scaled_pixel_images(self):
        return _to_numpy(self._results["scaled"])

    @staticmethod
    def _warp_pixel(pixel_space_point, spacing, origin, affine, camera_to_stage_affine=None):
        """Pixel -> physical -> camera-to-stage -> global affine (:2645-2683)."""
        p = pixel_space_point * spacing + origin
        if camera_to_stage_affine is not None:
            p = (np.asarray(camera_to_stage_affine) @ np.array([*list(p), 1]))[:-1]
        return (np.array(affine) @ np.array([*list(p), 1]))[:-1]

    @staticmethod
    def _warp_pixels(points, spacing, origin, affine, camera_to_stage_affine=None):
        """``_warp_pixel`` for every row of ``points`` (the reference loops in Python, :3134-3143).  Same
        operations in the same order per point - one float64 matrix-vector product per affine - with the
        array construction hoisted out of the loop, so the results are bit-identical."""
        phys = points * spacing + origin
        A = np.array(affine).astype(np.float64, copy=False)
        C = None if camera_to_stage_affine is None else np.asarray(camera_to_stage_affine).astype(np.float64, copy=False)
        h = np.ones((points.shape[0], 4), dtype=np.float64)
        h[:, :3] = phys
        # one einsum per affine instead of a Python loop of 4 x 4 matrix-vector products (0.12 s per 68 000 rows): each
        # output is the same four-term sum in the same order, so the rows are bit-identical to the loop's
        # (tests/test_pixeldecoder_host.py::test_warp_pixels_equals_the_per_point_loop holds that on 200 000 points)
        if C is not None:
            h[:, :3] = np.einsum("rc,nc->nr", C, h)[:, :3]
        return np.ascontiguousarray(np.einsum("rc,nc->nr", A, h)[:, :3])

    def _decoded_z_to_source_z(self, decoded_z):
        return float(self._z_range[0]) + decoded_z

    # ------------------------------------------------------- transcript extraction (:2908-3201)
    def _extract_barcodes(self, minimum_pixels: int = 3, maximum_pixels: int = 500, gpu_id: int = 0) -> None:
        """Connected components of equal codeword id -> transcript table (:2908-3201).

        Device: labelling + exact per-component sums.  Host (as in the reference, on a table
        with one row per component): size filters, means, inertia eigenvalues, coordinates.
        """
        torch = _torch()
        self._df_barcodes = pd.DataFrame()
        if self._optimize_normalization_weights:
            intensity = self._image_data_lp if self._filter_type == "lp" else self._image_data
        else:
            intensity = self._results["scaled"]
        B = int(intensity.shape[0])
        table = kernels.connected_component_table(self._results["decoded"], intensity,
                                                  self._results["magnitude"],
                                                  self._results["distance"], is_3d=self._is_3D)
        sums = table["sums"]
        area = sums[:, 0] if sums.size else np.zeros(0)
        max_size = max(int(minimum_pixels) - 1, 0)
        keep = (area <= maximum_pixels) & (area > max_size) & (table["code"] >= 0)
        label = np.nonzero(keep)[0] + 1  # scikit-image label = raster rank of the component
        sums, area = sums[keep], area[keep]
        code = table["code"][keep].astype(np.int32)
        n = int(area.size)
        z, y, x = (sums[:, 1] / area, sums[:, 2] / area, sums[:, 3] / area) if n else (np.zeros(0),) * 3
        if intensity.dtype == torch.float16:  # np.mean(float16): float32 sum -> float16 result
            means = (sums[:, kernels.SUM_COLS:].astype(np.float32) / area.astype(np.float32)[:, None]
                     ).astype(np.float16)
        else:
            means = (sums[:, kernels.SUM_COLS:] / area[:, None]).astype(np.float32)
        magnitude_mean = (sums[:, 10].astype(np.float32) / area.astype(np.float32)).astype(np.float16)
        # inertia tensor eigenvalues from central second moments (skimage inertia_tensor_eigvals)
        eig = np.zeros((n, 3))
        if n:
            mzz = sums[:, 4] / area - z * z
            myy = sums[:, 5] / area - y * y
            mxx = sums[:, 6] / area - x * x
            mzy = sums[:, 7] / area - z * y
            mzx = sums[:, 8] / area - z * x
            myx = sums[:, 9] / area - y * x
            T = np.zeros((n, 3, 3))
            T[:, 0, 0], T[:, 1, 1], T[:, 2, 2] = myy + mxx, mzz + mxx, mzz + myy
            T[:, 0, 1] = T[:, 1, 0] = -mzy
            T[:, 0, 2] = T[:, 2, 0] = -mzx
            T[:, 1, 2] = T[:, 2, 1] = -myx
            eig = np.clip(np.linalg.eigvalsh(T), 0, None)[:, ::-1]
        data = {"label": label, "area": area.astype(np.float64), "centroid-0": z, "centroid-1": y,
                "centroid-2": x}
        for k in range(B):
            data[f"intensity_mean-{k}"] = means[:, k]
        for k in range(3):
            data[f"inertia_tensor_eigvals-{k}"] = eig[:, k]
        data["distance_min"] = table["min_distance"][keep].astype(np.float32)
        data["magnitude_mean"] = magnitude_mean
        df = pd.DataFrame(data)
        df["decoded_id"] = code
        df = df[df["decoded_id"] >= 0].reset_index(drop=True)
        df["barcode_id"] = df["decoded_id"].astype(np.int32) + 1  # 1-based (:3088)
        df["gene_id"] = [self._gene_ids[i] for i in df["decoded_id"].to_numpy(dtype=np.int32)]
        df["tile_idx"] = self._tile_idx
        codebook_bool = self._codebook_matrix.astype(bool, copy=False)
        on_bits = np.argsort(~codebook_bool, axis=1, kind="stable")[:, :4].astype(np.int32) + 1
        on_sel = on_bits[df["decoded_id"].to_numpy(dtype=np.int32)]
        for k in range(4):
            df[f"on_bit_{k + 1}"] = on_sel[:, k] if len(df) else np.zeros(0, dtype=np.int32)
        df = df.rename(columns={"centroid-0": "z", "centroid-1": "y", "centroid-2": "x"})
        if self._z_crop:
            df["z"] = self._decoded_z_to_source_z(df["z"])
        df["tile_z"] = np.round(df["z"], 0).astype(int)
        df["tile_y"] = np.round(df["y"], 0).astype(int)
        df["tile_x"] = np.round(df["x"], 0).astype(int)
        pts = self._warp_pixels(df[["z", "y", "x"]].to_numpy(), self._spacing, self._origin, self._affine,
                                self._camera_to_stage_affine)
        df["global_z"] = np.round(pts[:, 0], 2)
        df["global_y"] = np.round(pts[:, 1], 2)
        df["global_x"] = np.round(pts[:, 2], 2)
        for i in range(1, self._n_merfish_bits + 1):
            df = df.rename(columns={f"intensity_mean-{i - 1}": f"bit{i:02d}_mean_intensity"})
        bit_cols = [f"bit{i:02d}_mean_intensity" for i in range(1, self._n_merfish_bits + 1)]
        bit_means = df[bit_cols].to_numpy(dtype=np.float64)
        total = bit_means.sum(axis=1)
        on0 = df[["on_bit_1", "on_bit_2", "on_bit_3", "on_bit_4"]].to_numpy(dtype=np.int32) - 1
        signal = np.take_along_axis(bit_means, on0, axis=1).sum(axis=1) if len(df) else np.zeros(0)
        df["signal_mean"] = signal / 4.0
        df["bkd_mean"] = (total - signal) / float(self._n_merfish_bits - 4)
        df["s-b_mean"] = df["signal_mean"] - df["bkd_mean"]
        df = df.drop(columns=["label", "decoded_id"])
        df = df[df["distance_min"] <= self._transcript_distance_threshold].reset_index(drop=True)
        self._df_barcodes = df.copy()

    def _save_barcodes(self) -> None:
        """Per-tile table to the datastore, or temp parquet while optimising (:3203-3233)."""
        if self._optimize_normalization_weights:
            self._temp_dir.mkdir(parents=True, exist_ok=True)
            path = self._temp_dir / ("tile" + str(self._tile_idx).zfill(3) + "_temp_decoded.parquet")
            self._df_barcodes.to_parquet(path)
        elif not self._barcodes_filtered:
            self._datastore.save_local_decoded_spots(self._df_barcodes, tile=self._tile_idx,
                                                     decode_run_key=self._decode_run_key)
        else:
            self._datastore.save_global_filtered_decoded_spots(self._df_filtered_barcodes,
                                                               decode_run_key=self._decode_run_key)

    @property
    def decoded_barcodes(self) -> pd.DataFrame:
        return self._df_barcodes.copy() if hasattr(self, "_df_barcodes") else pd.DataFrame()

    @property
    def decoded_image(self) -> np.ndarray:
        if not self._results:
            return np.empty((0,), dtype=np.int16)
        return self._decoded_image.copy()

    def save_decoded_barcodes(self) -> None:
        self._save_barcodes()

    def _prepare_normalization_state(self, normalization_method, use_normalization, gpu_id: int = 0,
                                     lowpass_sigma=DEFAULT_DECODE_LOWPASS_SIGMA) -> None:
        """iterative > global > none (:3274-3320)."""
        if normalization_method is None:
            normalization_method = "iterative" if use_normalization else "none"
        if normalization_method == "iterative":
            self._load_iterative_normalization_vectors(gpu_id=gpu_id)
        elif normalization_method == "global":
            self._iterative_normalization_loaded = False
            self._load_global_normalization_vectors(gpu_id=gpu_id, lowpass_sigma=lowpass_sigma)
        elif normalization_method == "none":
            self._iterative_normalization_loaded = False
            self._global_normalization_loaded = False
        else:
            raise ValueError("normalization_method must be one of 'iterative', 'global', "
                             f"'none', or None. Got {normalization_method!r}.")

    def _load_all_barcodes(self) -> None:
        """Gather per-tile tables (:3322-3384; the cell restriction needs shapely - out of scope)."""
        if self._optimize_normalization_weights:
            files = sorted(self._temp_dir.glob("*.parquet"), key=operator.attrgetter("name"))
            self._df_barcodes_loaded = pd.concat([pd.read_parquet(f) for f in files])
        elif self._load_tile_decoding:
            self._df_barcodes_loaded = pd.concat([
                self._datastore.load_local_decoded_spots(t, decode_run_key=self._decode_run_key)
                for t in self._datastore.tile_ids])
        else:
            self._df_filtered_barcodes = self._datastore.load_global_filtered_decoded_spots(
                decode_run_key=self._decode_run_key)
            self._df_barcodes_loaded = self._df_filtered_barcodes.copy()
            self._barcodes_filtered = True
        df = self._df_barcodes_loaded
        if df.empty:
            if "gene_id" not in df.columns:
                df["gene_id"] = pd.Series(dtype="string")
            for column in ("distance_min", "magnitude_mean", "area"):
                if column not in df.columns:
                    df[column] = pd.Series(dtype=np.float32)
        self._df_barcodes_loaded = df[df["gene_id"].notna() & df["gene_id"].astype(str).str.strip().ne("")]
        if "distance_min" not in self._df_barcodes_loaded.columns:
            raise ValueError("Decoded transcripts are missing 'distance_min'. Re-decode local "
                             "transcript parquet files with the exact two-threshold caller.")

    def _cleanup(self) -> None:
        """Drop the tile's HBM working set (:4426-4469)."""
        for name in ("_image_data_lp", "_image_data"):
            if hasattr(self, name):
                try:
                    delattr(self, name)
                except AttributeError:
                    pass
        self._results = {}
        self._df_barcodes = pd.DataFrame()
        gc.collect()
        torch = _torch()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            torch.cuda.empty_cache()

    def decode_one_tile(
        self,
        tile_idx: int = 0,
        gpu_id: int = 0,
        display_results: bool = False,
        return_results: bool = False,
        lowpass_sigma: Sequence[float] | None = DEFAULT_DECODE_LOWPASS_SIGMA,
        magnitude_threshold: Sequence[float] | None = None,
        minimum_pixels: float | None = None,
        use_normalization: bool | None = True,
        normalization_method: Literal["iterative", "global", "none"] | None = None,
        feature_predictor_threshold: float | None = 0.1,
    ):
        """Decode one tile (:4471-4579).  ``return_results`` gives the reference's 5-tuple of
        NumPy arrays ``(image, scaled f16, magnitude f16, distance f16, decoded int16)``."""
        torch = _torch()
        if display_results:
            raise NotImplementedError("napari display is a viewer feature and out of scope")
        with torch.cuda.device(gpu_id):
            if magnitude_threshold is None:
                magnitude_threshold = DEFAULT_DECODE_MAGNITUDE_THRESHOLD
            if minimum_pixels is None:
                minimum_pixels = self._default_minimum_pixels()
            self._prepare_normalization_state(normalization_method=normalization_method,
                                              use_normalization=use_normalization, gpu_id=gpu_id,
                                              lowpass_sigma=lowpass_sigma)
            self._tile_idx = tile_idx
            self._load_bit_data(feature_predictor_threshold=feature_predictor_threshold, gpu_id=gpu_id)
            self._filter_type = "raw"
            sigma = self._effective_lowpass_sigma(lowpass_sigma)
            if sigma is not None and not np.any(np.asarray(sigma, dtype=float) == 0):
                self._lp_filter(sigma=sigma, gpu_id=gpu_id)
            self._decode_pixels(magnitude_threshold=magnitude_threshold, gpu_id=gpu_id)
            self._extract_barcodes(minimum_pixels=minimum_pixels, gpu_id=gpu_id)
            if return_results:
                image = self._image_data_lp if self._filter_type == "lp" else self._image_data
                return (_to_numpy(image), self._scaled_pixel_images, self._magnitude_image,
                        self._distance_image, self._decoded_image)
        return None

    # --------------------------------------------------- normalisation by decoding (:4581-4757)
    def _run_optimize_iteration(self, tile_indices, gpu_id, temp_dir, iteration, lowpass_sigma,
                                magnitude_threshold, minimum_pixels, feature_predictor_threshold):
        """Body of ``_optimize_norm_worker`` (:386-410): global vectors seed iteration 0."""
        self._load_global_normalization_vectors(gpu_id=gpu_id, lowpass_sigma=lowpass_sigma)
        self._optimize_normalization_weights = True
        self._temp_dir = Path(temp_dir)
        use_norm = iteration > 0
        for tile_idx in tile_indices:
            self.decode_one_tile(tile_idx=tile_idx, display_results=False, return_results=False,
                                 lowpass_sigma=lowpass_sigma, magnitude_threshold=magnitude_threshold,
                                 minimum_pixels=minimum_pixels,
                                 feature_predictor_threshold=feature_predictor_threshold,
                                 normalization_method="iterative" if use_norm else "global",
                                 gpu_id=gpu_id)
            self._save_barcodes()

    def _shard(self, items):
        """This process's share: contiguous chunks over ``num_gpus`` (:4682, :4811-4820), or
        over ``torch.distributed`` ranks when the job runs one process per GPU."""
        if kernels._dist_ready():
            import torch.distributed as dist

            world, rank = dist.get_world_size(), dist.get_rank()
            chunk = (len(items) + world - 1) // world
            return [items[rank * chunk : min((rank + 1) * chunk, len(items))]], True
        chunk = (len(items) + self._num_gpus - 1) // self._num_gpus
        return [items[g * chunk : min((g + 1) * chunk, len(items))] for g in range(self._num_gpus)], False

    def optimize_normalization_by_decoding(
        self,
        n_random_tiles: int = 5,
        n_iterations: int = 10,
        minimum_pixels: float | None = None,
        feature_predictor_threshold: float | None = 0.1,
        lowpass_sigma: Sequence[float] | None = DEFAULT_DECODE_LOWPASS_SIGMA,
        magnitude_threshold: Sequence[float] | None = None,
        tile_indices: Sequence[int] | None = None,
        estimate_chromatic_affines: bool | None = None,
        excluded_gene_ids: Sequence[str] | None = None,
    ) -> None:
        """Refine normalisation vectors from exact-called transcripts (:4581-4757)."""
        if self._num_gpus < 1:
            raise RuntimeError("No GPUs allocated.")
        if estimate_chromatic_affines:
            raise NotImplementedError("chromatic-affine estimation is outside the hot path")
        if magnitude_threshold is None:
            magnitude_threshold = DEFAULT_DECODE_MAGNITUDE_THRESHOLD
        if minimum_pixels is None:
            minimum_pixels = self._default_minimum_pixels()
        self._optimization_excluded_gene_ids, excl = self._resolve_excluded_gene_ids(excluded_gene_ids)
        saved_exclusions = (self._excluded_gene_ids, self._excluded_codeword_indices)
        self._excluded_gene_ids, self._excluded_codeword_indices = self._optimization_excluded_gene_ids, excl
        blank = [g for g in self._optimization_excluded_gene_ids if g.lower().startswith("blank")]
        if blank:
            warnings.warn("Blank codewords already do not contribute to iterative normalization; "
                          "excluding them only suppresses their temporary assignments: "
                          + ", ".join(blank), stacklevel=2)
        all_tiles = list(range(len(self._datastore.tile_ids)))
        self._iterative_background_vector = None
        self._iterative_normalization_vector = None
        self._global_background_vector = None
        self._iterative_normalization_loaded = False
        self._optimize_normalization_weights = True
        self._load_global_normalization_vectors(gpu_id=0, recalculate=True, tile_indices=tile_indices,
                                                lowpass_sigma=lowpass_sigma)
        if self._decode_run_key is None:
            temp_dir = Path(tempfile.mkdtemp())
        else:
            temp_dir = self._datastore.decoded_temporary_dir(self._decode_run_key)
            temp_dir.mkdir(parents=True, exist_ok=True)
        self._temp_dir = temp_dir
        if tile_indices is not None:
            random_tiles = list(tile_indices)
        elif len(all_tiles) > n_random_tiles:
            random_tiles = sample(all_tiles, n_random_tiles)  # unseeded like the reference (:4679)
        else:
            random_tiles = all_tiles
        try:
            for iteration in range(n_iterations):
                for old in temp_dir.glob("*.parquet"):
                    old.unlink()
                shards, distributed = self._shard(random_tiles)
                if distributed or self._num_gpus == 1:
                    self._run_optimize_iteration(shards[0], 0 if not distributed else
                                                 _torch().cuda.current_device(), temp_dir, iteration,
                                                 lowpass_sigma, magnitude_threshold, minimum_pixels,
                                                 feature_predictor_threshold)
                    if distributed:
                        import torch.distributed as dist

                        dist.barrier()
                else:
                    procs = [_start_gpu_worker_process(
                        target=_optimize_norm_worker,
                        args=(self._datastore_path, subset, 0, self._n_merfish_bits, self._decode_mode,
                              temp_dir, iteration, lowpass_sigma, magnitude_threshold, minimum_pixels,
                              feature_predictor_threshold, False, self._optimization_excluded_gene_ids),
                        physical_gpu_id=gpu) for gpu, subset in enumerate(shards) if subset]
                    _join_gpu_workers(procs, "Iterative normalization")
                self._optimize_normalization_weights = True
                self._load_all_barcodes()
                if not self._is_3D:  # per-plane calls of one molecule collapse before they vote (:4734-4740)
                    voxel = self._datastore.voxel_size_zyx_um
                    self._remove_duplicates_within_tile(radius_xy=voxel[-1], radius_z=voxel[0])
                self._load_global_normalization_vectors(gpu_id=0, lowpass_sigma=lowpass_sigma)
                self._iterative_normalization_vectors(gpu_id=0)
        finally:
            self._excluded_gene_ids, self._excluded_codeword_indices = saved_exclusions
            self._cleanup()
            self._optimize_normalization_weights = False
            if self._decode_run_key is None:
                shutil.rmtree(temp_dir, ignore_errors=True)

    def decode_all_tiles(
        self,
        assign_to_cells: bool = True,
        lowpass_sigma: Sequence[float] | None = DEFAULT_DECODE_LOWPASS_SIGMA,
        magnitude_threshold: Sequence[float] | None = None,
        minimum_pixels: float | None = None,
        feature_predictor_threshold: float | None = 0.1,
        normalization_method: Literal["iterative", "global", "none"] = "iterative",
        duplicate_radius_xy: float | None = None,
        duplicate_radius_z: float | None = None,
        filter_method: Literal["blank_fraction", "lr"] = "blank_fraction",
        target_gross_misid_rate: float = 0.05,
        lr_fdr_target: float = 0.05,
    ) -> None:
        """Decode every tile, sharded over GPUs, then filter / de-duplicate / save the pooled
        transcript table on the host (:4759-4870).  Under ``torch.distributed`` rank 0 does the
        pooled post-processing after a barrier."""
        if self._num_gpus < 1:
            raise RuntimeError("No GPUs allocated.")
        if magnitude_threshold is None:
            magnitude_threshold = DEFAULT_DECODE_MAGNITUDE_THRESHOLD
        if minimum_pixels is None:
            minimum_pixels = self._default_minimum_pixels()
        self._validate_filter_configuration(filter_method=filter_method,
                                            target_gross_misid_rate=float(target_gross_misid_rate),
                                            lr_fdr_target=float(lr_fdr_target))
        all_tiles = list(range(len(self._datastore.tile_ids)))
        shards, distributed = self._shard(all_tiles)
        if distributed or self._num_gpus == 1:
            gpu = _torch().cuda.current_device() if distributed else 0
            self._optimize_normalization_weights = False
            mine = list(shards[0])
            prefetch = getattr(self._datastore, "prefetch_tile", None)  # disk-backed store: load the next
            for n, tile_idx in enumerate(mine):                          # tile's images under this tile's compute
                if prefetch is not None and n + 1 < len(mine):
                    prefetch(mine[n + 1], bits=self._datastore.bit_ids[: self._n_merfish_bits])
                self.decode_one_tile(tile_idx=tile_idx, gpu_id=gpu, lowpass_sigma=lowpass_sigma,
                                     magnitude_threshold=magnitude_threshold,
                                     minimum_pixels=minimum_pixels,
                                     feature_predictor_threshold=feature_predictor_threshold,
                                     normalization_method=normalization_method)
                self._save_barcodes()
                self._cleanup()
            if distributed:
                import torch.distributed as dist

                dist.barrier()
        else:
            procs = [_start_gpu_worker_process(
                target=decode_tiles_worker,
                args=(self._datastore_path, subset, 0, self._n_merfish_bits, self._verbose,
                      self._decode_mode, lowpass_sigma, magnitude_threshold, minimum_pixels,
                      feature_predictor_threshold, normalization_method),
                physical_gpu_id=gpu) for gpu, subset in enumerate(shards) if subset]
            _join_gpu_workers(procs, "Tile decoding")
        if distributed:
            import torch.distributed as dist

            if dist.get_rank() != 0:
                return
        self._load_tile_decoding = True
        self._load_all_barcodes()
        if self._verbose >= 1:
            print(f"Number of loaded barcodes: {len(self._df_barcodes_loaded)}")
            print(f"Verbosity:  {self._verbose}")
        self._apply_filter_method(filter_method=filter_method,
                                  target_gross_misid_rate=float(target_gross_misid_rate),
                                  lr_fdr_target=float(lr_fdr_target))
        if not self._is_3D:
            voxel = self._datastore.voxel_size_zyx_um
            self._remove_duplicates_within_tile(
                radius_xy=voxel[-1] if duplicate_radius_xy is None else float(duplicate_radius_xy),
                radius_z=voxel[0] if duplicate_radius_z is None else float(duplicate_radius_z))
        if len(all_tiles) > 1:
            self._remove_duplicates_in_tile_overlap()
        if assign_to_cells:
            self._assign_cells()
        self._save_barcodes()

    # ------------------------------------------------------------------------------------
    # post-decode filtering on the pooled table (host side)
    # ------------------------------------------------------------------------------------
    @staticmethod
    def _validate_filter_configuration(filter_method, target_gross_misid_rate: float, lr_fdr_target: float) -> None:
        """Each filter takes only its own target (:4872-4913)."""
        if filter_method == "blank_fraction":
            if lr_fdr_target != 0.05:
                raise ValueError("lr_fdr_target only applies when filter_method='lr'. "
                                 "Use target_gross_misid_rate with filter_method='blank_fraction'.")
            return
        if filter_method == "lr":
            if target_gross_misid_rate != 0.05:
                raise ValueError("target_gross_misid_rate only applies when filter_method='blank_fraction'. "
                                 "Use lr_fdr_target with filter_method='lr'.")
            return
        raise ValueError("filter_method must be one of 'blank_fraction' or 'lr'.")

    def _apply_filter_method(self, filter_method, target_gross_misid_rate: float, lr_fdr_target: float) -> None:
        """:4915-4950."""
        if self._verbose > 1:
            print(f"apply filter_method={filter_method}")
        if filter_method == "blank_fraction":
            self._filter_all_barcodes_blank_fraction(target_gross_misid_rate=float(target_gross_misid_rate))
            return
        if filter_method == "lr":
            self._filter_all_barcodes_LR(lr_fdr_target=float(lr_fdr_target))
            return
        raise ValueError("filter_method must be one of 'blank_fraction' or 'lr'.")

    def _filter_all_barcodes_blank_fraction(self, target_gross_misid_rate: float = 0.05, intensity_bins=None,
                                            voxel_number_bins=None, vector_distance_bins=None) -> None:
        """Blank-fraction histogram filter in (magnitude, area, distance) space (:3386-3846)."""
        from .postdecode import blank_fraction_filter

        filtered, diagnostics = blank_fraction_filter(
            self._df_barcodes_loaded, blank_count=self._blank_count, barcode_count=self._barcode_count,
            target_gross_misid_rate=target_gross_misid_rate, intensity_bins=intensity_bins,
            voxel_number_bins=voxel_number_bins, vector_distance_bins=vector_distance_bins)
        if self._verbose > 1:
            print("blank fraction filter diagnostics:")
            for key, value in diagnostics.items():
                if isinstance(value, (float, int, str, bool)):
                    print(f"{key}: {value}")
                else:
                    print(f"{key}: {type(value)} with shape {getattr(value, 'shape', 'N/A')}")
        self._blank_fraction_filter_results = diagnostics
        self._df_filtered_barcodes = filtered
        self._barcodes_filtered = True

    @staticmethod
    def _calculate_lr_fdr(df: pd.DataFrame, threshold: float, blank_count: int, barcode_count: int,
                          verbose: bool = False) -> float:
        """:3848-3905."""
        from .postdecode import lr_fdr

        value = lr_fdr(df, threshold, blank_count, barcode_count)
        if verbose > 1:
            print(f"threshold: {threshold}")
            print(f"lr_fdr: {value}")
        return value

    def _filter_all_barcodes_LR(self, lr_fdr_target: float = 0.05) -> None:
        """Logistic-regression blank classifier with an FDR target (:3907-4057)."""
        from .postdecode import lr_filter

        filtered, threshold = lr_filter(self._df_barcodes_loaded, is_3d=self._is_3D, blank_count=self._blank_count,
                                        barcode_count=self._barcode_count, lr_fdr_target=lr_fdr_target)
        if threshold is None and self._verbose >= 1:
            print("Insufficient Blank barcodes called for filtering.")
        self._df_filtered_barcodes = filtered
        self._barcodes_filtered = True

    def _remove_duplicates_in_tile_overlap(self, radius: float = 0.75) -> None:
        """:4137-4177."""
        from .postdecode import remove_duplicates_in_tile_overlap

        self._df_filtered_barcodes, dropped = remove_duplicates_in_tile_overlap(self._df_filtered_barcodes, radius)
        if self._verbose > 1:
            print("Average distance_min of dropped points (overlap): " + str(dropped.mean() if dropped.size else 0))
            print("Dropped points: " + str(len(dropped)))

    def _remove_duplicates_within_tile(self, radius_xy: float = 0.1, radius_z: float = 0.50) -> None:
        """:4179-4363 - acts on the filtered table when there is one, else on the loaded table."""
        from .postdecode import remove_duplicates_within_tile

        filtered = getattr(self, "_df_filtered_barcodes", None) is not None
        df = self._df_filtered_barcodes if filtered else getattr(self, "_df_barcodes_loaded", None)
        if df is None or df.empty or len(df) < 2:
            return
        dmin = df["distance_min"].to_numpy(dtype=float)
        out, dropped = remove_duplicates_within_tile(df, radius_xy=radius_xy, radius_z=radius_z)
        if self._verbose > 1:
            avg = float(dmin[dropped].mean()) if dropped else 0.0
            print("Average distance_min of dropped points (within-tile, same gene, clusters): " + str(avg))
            print("Dropped points: " + str(len(dropped)))
        if filtered:
            self._df_filtered_barcodes = out.copy()
        else:
            self._df_barcodes_loaded = out.copy()

    def _assign_cells(self) -> None:
        """Cell assignment from Cellpose ROIs needs roifile / shapely / rtree and a segmentation on
        disk (:4076-4135); without them the reference prints and returns, and so does this."""
        try:
            from roifile import roiread  # noqa: F401
        except ImportError as exc:
            print(f"Failed to read ROIs: {exc}")
            return
        raise NotImplementedError("cell assignment from ImageJ ROIs is outside the hot path")

    def optimize_filtering(self, assign_to_cells: bool = True, duplicate_radius_xy: float | None = None,
                           duplicate_radius_z: float | None = None,
                           filter_method: Literal["blank_fraction", "lr"] = "blank_fraction",
                           target_gross_misid_rate: float = 0.05, lr_fdr_target: float = 0.05) -> None:
        """Re-apply the downstream filter to previously decoded tiles (:4952-5030)."""
        if self._verbose >= 1:
            print("reprocess existing: load decoded transcripts")
        self._load_tile_decoding = True
        self._load_all_barcodes()
        if self._verbose >= 1:
            print(f"Number of loaded barcodes: {len(self._df_barcodes_loaded)}")
        self._load_tile_decoding = False
        self._validate_filter_configuration(filter_method=filter_method,
                                            target_gross_misid_rate=float(target_gross_misid_rate),
                                            lr_fdr_target=float(lr_fdr_target))
        all_tiles = list(range(len(self._datastore.tile_ids)))
        if self._verbose != 0:
            self._verbose = 2
        self._apply_filter_method(filter_method=filter_method, target_gross_misid_rate=float(target_gross_misid_rate),
                                  lr_fdr_target=float(lr_fdr_target))
        if len(all_tiles) or not self._is_3D:
            if not self._is_3D:
                voxel = self._datastore.voxel_size_zyx_um
                self._remove_duplicates_within_tile(
                    radius_xy=voxel[-1] if duplicate_radius_xy is None else float(duplicate_radius_xy),
                    radius_z=voxel[0] if duplicate_radius_z is None else float(duplicate_radius_z))
            else:
                self._remove_duplicates_in_tile_overlap()
        if self._verbose != 0:
            self._verbose = 1
        if assign_to_cells:
            self._assign_cells()
        self._save_barcodes()
        if self._verbose >= 1:
            print(f"Number of retained barcodes: {len(self._df_filtered_barcodes)}")

