"""The hot path's view of a qi2lab datastore directory: images *and* sidecar metadata on disk.

``Qi2labDiskStore`` opens (or creates) a directory laid out like the reference's ``qi2labDataStore``
(qi2labDataStore.py) and serves the methods the tile operators call from the same
files the reference writes:

* ``datastore_state.json``; ``calibrations/attributes.json`` - experiment constants, codebook, experiment
  order, chromatic affines, default and run-scoped normalisation vectors (:1129-1271);
* ``fiducial/<tile>/<round>/attributes.json`` - ``bit_linker``, ``excitation_um`` / ``emission_um``,
  ``stage_zyx_um`` + ``affine_zyx_px``, ``local_round_transform_zyx_um``, and on the first round the tile's
  ``affine_zyx_um`` / ``origin_zyx_um`` / ``spacing_zyx_um`` (:3209-3350, 3901-4030, 5137-5240);
* ``readouts/<tile>/<bit>/attributes.json`` - ``round_linker``, wavelengths (:3083-3207, 3352-3530);
* ``decoded[/<run>]/<tile>_decoded_features.parquet``, ``all_tiles_filtered_decoded_features[/<run>]/
  decoded_features.{parquet,csv.gz}``, ``<predictor>_localizations/<tile>/<bit>.parquet`` (:5347-5520);
* every image as an OME-Zarr store through the I/O fast path (``ZarrImageDataStore`` / ``zarr_io``), the
  channel PSFs among them (``calibrations/psf_data/psf_NNN.ome.zarr`` + ``psf_manifest``, :698-745).

Key names, value shapes, defaults and the print-and-return-``None`` convention for missing data follow the
reference; ``tests/golden/qi2lab_sidecars.json`` holds a tree written and read back by the reference's own class
(``tests/golden/make_golden_datastore.py``) and ``tests/test_datastore_sidecars.py`` checks both directions.
Not covered (outside the hot path): shading / noise-map calibration images, fused and segmentation products,
store validation.
"""

from __future__ import annotations

import json
import re
from pathlib import Path
from typing import Any, Mapping

import numpy as np
import pandas as pd

from .datastore import ZarrImageDataStore
from . import zarr_io

_REQUIRED = ("num_rounds", "num_tiles", "num_bits", "codebook", "voxel_size_zyx_um", "microscope_type")
_DEFAULT_STATE = {"Version": 0.6, "Initialized": True, "Calibrations": False, "Corrected": False,
                  "LocalRegistered": False, "GlobalRegistered": False, "Fused": False, "SegmentedCells": False,
                  "DecodedSpots": False, "FilteredSpots": False}


def _jsonable(value: Any) -> Any:
    return zarr_io._json_compatible(value)


class Qi2labDiskStore(ZarrImageDataStore):
    in_memory = False

    def __init__(self, datastore_path, *, device=None, num_threads=None, write_behind=False, prefetch_threads=None,
                 write_threads=None, max_pending_writes=6, **create):
        """Open ``datastore_path`` if it holds ``calibrations/attributes.json``; otherwise create it from
        ``create`` = ``num_tiles, num_rounds, num_bits, codebook`` (DataFrame: gene_id + one column per bit)
        [, ``voxel_size_zyx_um``, ``microscope_type``, ``experiment_order`` (rows: fiducial round, bits),
        ``channels_in_data`` ...]."""
        root = Path(datastore_path)
        calib = root / "calibrations" / "attributes.json"
        if calib.exists():
            if create:
                raise ValueError("the datastore exists: open it without creation arguments")
            attrs = json.loads(calib.read_text())
            missing = [k for k in _REQUIRED if k not in attrs]
            if missing:
                raise KeyError(f"Calibration attributes incomplete: {missing}")
            rows = attrs["codebook"]
            ncol = len(rows[0]) if rows else 0
            codebook = pd.DataFrame(rows, columns=["gene_id"] + [f"bit{i:02d}" for i in range(1, ncol)])
            kwargs = dict(num_tiles=int(attrs["num_tiles"]), num_rounds=int(attrs["num_rounds"]),
                          num_bits=int(attrs["num_bits"]), codebook=codebook,
                          voxel_size_zyx_um=attrs["voxel_size_zyx_um"], microscope_type=attrs["microscope_type"])
        else:
            kwargs = {k: create.pop(k) for k in ("num_tiles", "num_rounds", "num_bits", "codebook")}
            kwargs["voxel_size_zyx_um"] = create.pop("voxel_size_zyx_um", (0.315, 0.098, 0.098))
            kwargs["microscope_type"] = create.pop("microscope_type", "3D")
        super().__init__(root, device=device, num_threads=num_threads, write_behind=write_behind,
                         prefetch_threads=prefetch_threads, write_threads=write_threads,
                         max_pending_writes=max_pending_writes, **kwargs)
        self._attr_cache = {}
        self._calibrations_zarr_path = root / "calibrations"
        self._decoded_root_path = root / "decoded"
        self._segmentation_root_path = root / "segmentation"
        for p in (self._calibrations_zarr_path, self._decoded_root_path, root / "fused", self._segmentation_root_path):
            p.mkdir(parents=True, exist_ok=True)
        if not calib.exists():
            self._init_calibrations(kwargs, create)

    # ------------------------------------------------------------------ json plumbing
    @staticmethod
    def _load_from_json(path: Path) -> dict:
        try:
            with open(path, encoding="utf-8") as fh:
                return json.load(fh)
        except (FileNotFoundError, json.JSONDecodeError):
            return {}

    @staticmethod
    def _save_to_json(dictionary: Mapping, path: Path) -> None:
        path.parent.mkdir(parents=True, exist_ok=True)
        with open(path, "w", encoding="utf-8") as fh:
            json.dump(dictionary, fh, indent=4)

    def _calibrations_attributes_path(self) -> Path:
        return self._calibrations_zarr_path / "attributes.json"

    def _load_calibrations_attributes(self) -> dict:
        return self._load_from_json(self._calibrations_attributes_path())

    def _set_calibration_attribute(self, key: str, value: Any) -> None:
        attrs = self._load_calibrations_attributes()
        attrs[str(key)] = _jsonable(value)
        self._save_to_json(attrs, self._calibrations_attributes_path())

    def _init_calibrations(self, kwargs, extra) -> None:
        codebook = kwargs["codebook"]
        attrs = {
            "num_rounds": int(kwargs["num_rounds"]), "num_tiles": int(kwargs["num_tiles"]),
            "num_bits": int(kwargs["num_bits"]), "microscope_type": str(kwargs["microscope_type"]),
            "voxel_size_zyx_um": [float(v) for v in kwargs["voxel_size_zyx_um"]],
            "codebook": _jsonable(codebook.values),
        }
        order = extra.pop("experiment_order", None)
        if order is not None:
            attrs["exp_order"] = _jsonable(np.asarray(order))
        for key in ("channels_in_data", "camera_model", "tile_overlap", "binning", "na", "ri", "e_per_ADU"):
            if key in extra:
                attrs[key] = _jsonable(extra.pop(key))
        if extra:
            raise TypeError(f"unknown creation arguments: {sorted(extra)}")
        self._save_to_json(attrs, self._calibrations_attributes_path())
        self._save_to_json(dict(_DEFAULT_STATE), self._datastore_path / "datastore_state.json")
        # initialize_tile (:2840-2960): one folder + sidecar per entity, linkers from the experiment order
        order = None if order is None else np.asarray(order, dtype=np.int64)
        for tile in self.tile_ids:
            for r, rid in enumerate(self.round_ids):
                side = {} if order is None else {"bit_linker": order[r, 1:].astype(int).tolist()}
                self._save_to_json(side, self._entity_attributes_path(self._fiducial_root_path / tile / rid))
            for b, bid in enumerate(self.bit_ids):
                side = {}
                if order is not None:
                    rows = np.nonzero((order[:, 1:] == b + 1).any(axis=1))[0]
                    if rows.size:
                        side["round_linker"] = int(order[rows[0], 0])
                self._save_to_json(side, self._entity_attributes_path(self._readouts_root_path / tile / bid))

    @staticmethod
    def _entity_attributes_path(entity_root: Path) -> Path:
        return Path(entity_root) / "attributes.json"

    def _load_entity_attributes(self, entity_root: Path) -> dict:
        """Image-level attributes of the entity's stores, overlaid by its sidecar (:1860-1902)."""
        merged: dict[str, Any] = {}
        for name in ("corrected_data", "decon_data", f"{self.feature_predictor_folder_name}_data",
                     "opticalflow_xform_px"):
            store = zarr_io.image_store_path(Path(entity_root) / name)
            if (store / "zarr.json").exists():
                merged.update(zarr_io.read_extra_attributes(store))
        merged.update(self._load_from_json(self._entity_attributes_path(entity_root)))
        return merged

    def _save_entity_attributes(self, entity_root: Path, updates: Mapping[str, Any]) -> None:
        if not updates:
            return
        path = self._entity_attributes_path(entity_root)
        attrs = self._load_from_json(path)
        attrs.update({str(k): _jsonable(v) for k, v in dict(updates).items()})
        self._save_to_json(attrs, path)

    def _entity(self, tile, round=None, bit=None):
        k = self._key(tile, round, bit)
        if k is None:
            return None
        return (self._fiducial_root_path if k[1] == "round" else self._readouts_root_path) / k[0] / k[2]

    # ------------------------------------------------------------------ store-level properties
    @property
    def datastore_state(self):
        return self._load_from_json(self._datastore_path / "datastore_state.json") or None

    @datastore_state.setter
    def datastore_state(self, value):
        if value:  # the in-memory base class assigns {} during construction
            self._save_to_json(dict(value), self._datastore_path / "datastore_state.json")

    @property
    def num_rounds(self):
        return len(self.round_ids)

    @property
    def channel_psfs(self):
        """PSFs of ``calibrations/psf_data/psf_NNN.ome.zarr`` in manifest order: one stacked array when they
        share a shape, else a list (:677-696, 2518-2540); None when the store has none."""
        psfs = getattr(self, "_psfs", None)
        if psfs is None:
            manifest = self._load_calibrations_attributes().get("psf_manifest")
            root = self._datastore_path / "calibrations" / "psf_data"
            if isinstance(manifest, dict) and manifest:
                ids = [manifest[k]["id"] for k in sorted(manifest, key=int)]
            elif root.exists():
                ids = sorted(p.name[: -len(".ome.zarr")] for p in root.glob("psf_*.ome.zarr"))
            else:
                ids = []
            if not ids:
                return None
            psfs = self._psfs = [np.asarray(self._io.read_image(root / i), dtype=np.float32) for i in ids]
        if len({tuple(p.shape) for p in psfs}) == 1:
            return np.stack(psfs, axis=0)
        return psfs

    @channel_psfs.setter
    def channel_psfs(self, value):
        if value is None:      # the in-memory base class initialises the attribute
            return
        if isinstance(value, np.ndarray) and value.dtype != object:
            psf_list = [np.asarray(value[i], dtype=np.float32) for i in range(value.shape[0])] if value.ndim >= 3 \
                else [np.asarray(value, dtype=np.float32)]
        else:
            psf_list = [np.asarray(psf, dtype=np.float32) for psf in list(value)]
        if not psf_list:
            raise ValueError("channel_psfs cannot be empty.")
        self._psfs = psf_list
        root = self._datastore_path / "calibrations" / "psf_data"
        manifest = {}
        for idx, psf in enumerate(psf_list):
            self._io.write_image(root / f"psf_{idx:03d}", psf)
            manifest[str(idx)] = {"id": f"psf_{idx:03d}", "shape_zyx": list(psf.shape)}
        self._set_calibration_attribute("psf_manifest", manifest)

    def load_chromatic_affine_transforms_zyx_um(self) -> dict:
        calibration = self._load_calibrations_attributes().get("chromatic_affine_transforms_zyx_um", {})
        return calibration if isinstance(calibration, dict) else {}

    def save_chromatic_affine_transforms_zyx_um(self, calibration: Mapping[str, Any]) -> None:
        self._set_calibration_attribute("chromatic_affine_transforms_zyx_um", calibration)

    def load_chromatic_affine_transform_zyx_um(self, channel_name=None, channel_index=None, wavelength_um=None):
        """One channel's 4x4 physical affine, identity when absent (:148-232)."""
        channels = self.load_chromatic_affine_transforms_zyx_um().get("channels", {})
        if not isinstance(channels, Mapping):
            return np.eye(4, dtype=np.float32)
        candidates = []
        if channel_name is not None and isinstance(channels.get(str(channel_name)), Mapping):
            candidates.append(channels[str(channel_name)])
        if channel_index is not None:
            candidates += [c for c in channels.values()
                           if isinstance(c, Mapping) and int(c.get("channel_index", -1)) == int(channel_index)]
        if wavelength_um is not None:
            candidates += [c for c in channels.values() if isinstance(c, Mapping) and c.get("wavelength_um") is not None
                           and np.isclose(float(c["wavelength_um"]), float(wavelength_um))]
        for channel in candidates:
            if channel.get("affine_zyx_um") is not None:
                return np.asarray(channel["affine_zyx_um"], dtype=np.float32)
        return np.eye(4, dtype=np.float32)

    def set_chromatic_affine_transform_zyx_um(self, wavelength_um, transform):
        calibration = self.load_chromatic_affine_transforms_zyx_um()
        channels = dict(calibration.get("channels", {}))
        name = f"{float(wavelength_um):.4f}um"
        channels[name] = {"channel_index": len(channels), "wavelength_um": float(wavelength_um),
                          "affine_zyx_um": np.asarray(transform, dtype=np.float32).tolist()}
        calibration["channels"] = channels
        self.save_chromatic_affine_transforms_zyx_um(calibration)

    # ------------------------------------------------------------------ per-entity metadata
    def load_local_round_linker(self, tile, bit):
        root = self._entity(tile, bit=bit)
        if root is None:
            return None
        value = self._load_entity_attributes(root).get("round_linker")
        if value is None:
            print("Round linker attribute not found.")
            return None
        return int(value)

    def save_local_round_linker(self, round_linker, tile, bit):
        root = self._entity(tile, bit=bit)
        if root is not None:
            self._save_entity_attributes(root, {"round_linker": int(round_linker)})

    def set_round_linker(self, tile, bit, round_1based: int):
        self.save_local_round_linker(round_1based, tile, bit)

    def load_local_bit_linker(self, tile, round):
        root = self._entity(tile, round=round)
        if root is None:
            return None
        value = self._load_entity_attributes(root).get("bit_linker")
        if value is None:
            print("Bit linker attribute not found.")
            return None
        return [int(v) for v in value]

    def save_local_bit_linker(self, bit_linker, tile, round):
        root = self._entity(tile, round=round)
        if root is not None:
            self._save_entity_attributes(root, {"bit_linker": [int(v) for v in bit_linker]})

    def load_local_wavelengths_um(self, tile, round=None, bit=None):
        root = self._entity(tile, round, bit)
        if root is None:
            return None
        attrs = self._load_entity_attributes(root)
        try:
            return (attrs["excitation_um"], attrs["emission_um"])
        except KeyError:
            print("Wavelength attributes not found.")
            return None

    def save_local_wavelengths_um(self, wavelengths_um, tile, round=None, bit=None):
        root = self._entity(tile, round, bit)
        if root is not None:
            self._save_entity_attributes(root, {"excitation_um": float(wavelengths_um[0]),
                                                "emission_um": float(wavelengths_um[1])})

    def set_corrected_image(self, image, tile, round=None, bit=None, wavelengths_um=(0.561, 0.590)):
        k = self._key(tile, round, bit)
        self._corrected[k] = image
        self.save_local_wavelengths_um(wavelengths_um, tile, round=round, bit=bit)

    def save_local_corrected_image(self, image, tile, gain_correction: bool = True, hotpixel_correction: bool = True,
                                   shading_correction: bool = False, psf_idx: int = 0, round=None, bit=None,
                                   return_future=False):
        """:3639-3770 - the image store plus the correction flags and PSF index in the entity's sidecar."""
        k = self._key(tile, round, bit)
        root = self._entity(tile, round, bit)
        if k is None or root is None:
            return None
        self._corrected[k] = image
        self._save_entity_attributes(root, {"gain_correction": bool(gain_correction),
                                            "hotpixel_correction": bool(hotpixel_correction),
                                            "shading_correction": bool(shading_correction), "psf_idx": int(psf_idx)})

    def load_local_round_transform_zyx_um(self, tile, round):
        root = self._entity(tile, round=round)
        if root is None:
            return None
        value = self._load_entity_attributes(root).get("local_round_transform_zyx_um")
        if value is None:
            print("Local round transform mapping back to first round not found.")
            return None
        return np.asarray(value, dtype=np.float32)

    def save_local_round_transform_zyx_um(self, transform_zyx_um, tile, round):
        root = self._entity(tile, round=round)
        if root is not None:
            self._save_entity_attributes(
                root, {"local_round_transform_zyx_um": np.asarray(transform_zyx_um, dtype=np.float32).tolist()})

    def load_local_stage_position_zyx_um(self, tile, round=0):
        root = self._entity(tile, round=round)
        if root is None:
            return None
        attrs = self._load_entity_attributes(root)
        stage, affine = attrs.get("stage_zyx_um"), attrs.get("affine_zyx_px")
        if stage is None or affine is None:
            print("Stage position attribute not found.")
            return None
        return np.asarray(stage, dtype=np.float32), np.asarray(affine, dtype=np.float32)

    def save_local_stage_position_zyx_um(self, stage_zyx_um, affine_zyx_px, tile, round):
        root = self._entity(tile, round=round)
        if root is not None:
            self._save_entity_attributes(root, {"stage_zyx_um": np.asarray(stage_zyx_um, dtype=np.float32).tolist(),
                                                "affine_zyx_px": np.asarray(affine_zyx_px, dtype=np.float32).tolist()})

    def set_stage_position_zyx_um(self, tile, origin, camera_to_stage=None, round=0):
        c2s = np.eye(4, dtype=np.float32) if camera_to_stage is None else camera_to_stage
        self.save_local_stage_position_zyx_um(origin, c2s, tile, round)

    def load_global_coord_xforms_um(self, tile):
        root = self._entity(tile, round=0)
        if root is None:
            return None, None, None
        attrs = self._load_entity_attributes(root)
        try:
            return (np.asarray(attrs["affine_zyx_um"], dtype=np.float32),
                    np.asarray(attrs["origin_zyx_um"], dtype=np.float32),
                    np.asarray(attrs["spacing_zyx_um"], dtype=np.float32))
        except KeyError:
            print("Global coordinate transforms not found")
            return None, None, None

    def save_global_coord_xforms_um(self, affine_zyx_um, origin_zyx_um, spacing_zyx_um, tile):
        root = self._entity(tile, round=0)
        if root is not None:
            self._save_entity_attributes(root, {
                "affine_zyx_um": np.asarray(affine_zyx_um, dtype=np.float32).tolist(),
                "origin_zyx_um": np.asarray(origin_zyx_um, dtype=np.float32).tolist(),
                "spacing_zyx_um": np.asarray(spacing_zyx_um, dtype=np.float32).tolist()})

    def set_global_coord_xforms_um(self, tile, affine, origin, spacing):
        self.save_global_coord_xforms_um(affine, origin, spacing, tile)

    # ------------------------------------------------------------------ decode products
    @staticmethod
    def _validate_decode_run_key(decode_run_key):
        if decode_run_key is None:
            return None
        decode_run_key = str(decode_run_key)
        if not re.fullmatch(r"[A-Za-z0-9_.-]+", decode_run_key):
            raise ValueError("decode_run_key may only contain letters, numbers, '.', '_', and '-'.")
        return decode_run_key

    def _decoded_run_root(self, decode_run_key=None) -> Path:
        key = self._validate_decode_run_key(decode_run_key)
        return self._decoded_root_path if key is None else self._decoded_root_path / key

    def decoded_temporary_dir(self, decode_run_key=None, iteration=None) -> Path:
        root = self._decoded_run_root(decode_run_key) / "temporary"
        return root if iteration is None else root / f"iteration_{int(iteration):03d}"

    def _global_filtered_decoded_root(self, decode_run_key=None) -> Path:
        root = self._datastore_path / "all_tiles_filtered_decoded_features"
        key = self._validate_decode_run_key(decode_run_key)
        return root if key is None else root / key

    def save_decode_normalization_vectors(self, decode_run_key, kind, normalization, background,
                                          decode_mode=None, metadata=None):
        if kind not in ("global", "iterative"):
            raise ValueError("kind must be one of 'global' or 'iterative'.")
        norm = np.asarray(normalization, dtype=np.float32)
        bkd = np.asarray(background, dtype=np.float32)
        attrs = self._load_calibrations_attributes()
        if decode_run_key is None:
            attrs[f"{kind}_normalization_vector"] = norm.tolist()
            attrs[f"{kind}_background_vector"] = bkd.tolist()
            if metadata is not None:
                all_meta = dict(attrs.get("decode_normalization_metadata", {}))
                all_meta[kind] = _jsonable(dict(metadata))
                attrs["decode_normalization_metadata"] = all_meta
        else:
            key = self._validate_decode_run_key(decode_run_key)
            runs = dict(attrs.get("decode_normalization_runs", {}))
            run = dict(runs.get(key, {}))
            if decode_mode is not None:
                run["decode_mode"] = str(decode_mode)
            if metadata is not None:
                run[f"{kind}_metadata"] = _jsonable(dict(metadata))
            run[f"{kind}_normalization_vector"] = norm.tolist()
            run[f"{kind}_background_vector"] = bkd.tolist()
            runs[key] = run
            attrs["decode_normalization_runs"] = runs
        self._save_to_json(attrs, self._calibrations_attributes_path())

    def load_decode_normalization_vectors(self, decode_run_key, kind):
        if kind not in ("global", "iterative"):
            raise ValueError("kind must be one of 'global' or 'iterative'.")
        attrs = self._load_calibrations_attributes()
        if decode_run_key is not None:
            attrs = attrs.get("decode_normalization_runs", {}).get(self._validate_decode_run_key(decode_run_key), {})
        norm, bkd = attrs.get(f"{kind}_normalization_vector"), attrs.get(f"{kind}_background_vector")
        if decode_run_key is None:   # the default vectors are two independent properties (:879-1030)
            return (None if norm is None else np.asarray(norm, dtype=np.float32),
                    None if bkd is None else np.asarray(bkd, dtype=np.float32))
        if norm is None or bkd is None:
            return None, None
        return np.asarray(norm, dtype=np.float32), np.asarray(bkd, dtype=np.float32)

    def load_decode_normalization_metadata(self, decode_run_key, kind):
        if kind not in ("global", "iterative"):
            raise ValueError("kind must be one of 'global' or 'iterative'.")
        attrs = self._load_calibrations_attributes()
        if decode_run_key is None:
            meta = attrs.get("decode_normalization_metadata", {}).get(kind)
        else:
            meta = attrs.get("decode_normalization_runs", {}).get(
                self._validate_decode_run_key(decode_run_key), {}).get(f"{kind}_metadata")
        return dict(meta) if isinstance(meta, Mapping) else None

    def save_local_decoded_spots(self, df, tile, decode_run_key=None):
        t = self._tile(tile)
        if t is None:
            print("set valid tiled id")
            return None
        root = self._decoded_run_root(decode_run_key)
        root.mkdir(parents=True, exist_ok=True)
        self._save_to_parquet(df, root / f"{t}_decoded_features.parquet")

    def load_local_decoded_spots(self, tile, decode_run_key=None):
        t = self._tile(tile)
        if t is None:
            print("set valid tiled id")
            return None
        path = self._decoded_run_root(decode_run_key) / f"{t}_decoded_features.parquet"
        if not path.exists():
            print("Decoded spots not found.")
            return None
        return pd.read_parquet(path)

    def save_global_filtered_decoded_spots(self, df, decode_run_key=None):
        root = self._global_filtered_decoded_root(decode_run_key)
        root.mkdir(parents=True, exist_ok=True)
        self._save_to_parquet(df, root / "decoded_features.parquet")
        df.to_csv(root / "decoded_features.csv.gz", index=False, compression="gzip")

    def load_global_filtered_decoded_spots(self, decode_run_key=None):
        path = self._global_filtered_decoded_root(decode_run_key) / "decoded_features.parquet"
        if not path.exists():
            print("Global, filtered, decoded spots not found.")
            return None
        return pd.read_parquet(path)

    @staticmethod
    def _save_to_parquet(df, path) -> None:
        df.to_parquet(path, index=False)
