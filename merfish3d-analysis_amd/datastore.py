"""Datastores for the tile operators: the qi2labDataStore duck-type at three levels.

* ``InMemoryDataStore`` - every method the operators call, on dictionaries: tests and ``bench.py`` run
  without touching a disk (the role the ``SimpleNamespace`` / dataclass fakes play in the reference's own
  tests, ``tests/test_chromatic_estimator.py:9-27``).  Arrays may be NumPy arrays or torch tensors already in
  HBM (then nothing crosses PCIe when an operator loads them).
* ``ZarrImageDataStore`` - the same with every image on disk as an OME-Zarr store in the reference's folder
  layout, read and written through the I/O fast path (``zarr_io``), with next-tile read-ahead.
* ``qi2lab_store.Qi2labDiskStore`` (its own module) - additionally the reference's JSON sidecars and parquet
  tables, i.e. a directory the reference's ``qi2labDataStore`` class can open and vice versa.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import pandas as pd


class _Ready:
    """Mimics the tensorstore future returned with ``return_future=True`` (:2264-2267)."""

    def __init__(self, value):
        self._value = value

    def result(self):
        return self._value


class InMemoryDataStore:
    in_memory = True  # children of a spawn-per-GPU run would write into their own copies

    def __init__(self, *, num_tiles: int, num_rounds: int, num_bits: int, codebook: pd.DataFrame,
                 voxel_size_zyx_um=(0.315, 0.098, 0.098), microscope_type: str = "3D",
                 datastore_path: str | Path = "/tmp/m3d_inmemory_datastore", **experiment):
        # experiment constants of qi2labDataStore (experiment_order, channels_in_data, camera_model, tile_overlap,
        # binning, na, ri, e_per_ADU): kept as attributes, the disk store writes them to calibrations/attributes.json
        unknown = set(experiment) - {"experiment_order", "channels_in_data", "camera_model", "tile_overlap", "binning",
                                     "na", "ri", "e_per_ADU"}
        if unknown:
            raise TypeError(f"unknown creation arguments: {sorted(unknown)}")
        for key, value in experiment.items():
            setattr(self, key, value)
        self.tile_ids = [f"tile{i:04d}" for i in range(num_tiles)]
        self.round_ids = [f"round{i + 1:03d}" for i in range(num_rounds)]
        self.bit_ids = [f"bit{i + 1:03d}" for i in range(num_bits)]
        self.codebook = codebook
        self.voxel_size_zyx_um = tuple(float(v) for v in voxel_size_zyx_um)
        self.microscope_type = microscope_type
        self.channel_psfs = None
        self.datastore_state = {}
        self._datastore_path = Path(datastore_path)
        self._segmentation_root_path = self._datastore_path / "segmentation"  # qi2labDataStore.py: a folder, maybe empty
        self.feature_predictor_folder_name = "feature_predictor"  # qi2labDataStore.py:1322
        self._corrected = {}      # (tile, "round"/"bit", id) -> array
        self._fiducial_decon = {}
        self._readout_decon = {}
        self._predictor = {}
        self._round_linker = {}   # (tile, bit) -> 1-based round
        self._wavelengths = {}    # (tile, kind, id) -> (ex, em)
        self._round_xform = {}
        self._flow = {}
        self._stage = {}
        self._global_xform = {}
        self._chromatic = {}
        self._norm_vectors = {}
        self._norm_meta = {}
        self._norm_decode_mode = {}
        self._decoded = {}
        self._filtered = {}
        self._predictor_spots = {}
        order = experiment.get("experiment_order")
        if order is not None:   # initialize_tile (qi2labDataStore.py:2840-2960): bit <-> round links from the order
            order = np.asarray(order, dtype=np.int64)
            for tile in self.tile_ids:
                for b, bit_id in enumerate(self.bit_ids):
                    rows = np.nonzero((order[:, 1:] == b + 1).any(axis=1))[0]
                    if rows.size:
                        self._round_linker[(tile, bit_id)] = int(order[rows[0], 0])

    def load_codebook_parsed(self):
        """``(gene_ids, codebook_matrix)`` (qi2labDataStore.py:2841-2868; the simulation scorer's CLI reads it,
        cli/statphysbio_simulation/calculate_F1.py:230)."""
        try:
            gene_ids = self.codebook.iloc[:, 0].tolist()
            return gene_ids, self.codebook.iloc[:, 1:].to_numpy().astype(int)
        except (AttributeError, KeyError, ValueError, TypeError):
            print("Error parsing codebook.")
            return None

    # ---- id helpers (qi2labDataStore.py:2501-2515) ------------------------------------
    @property
    def num_tiles(self):
        return len(self.tile_ids)

    @property
    def num_bits(self):
        return len(self.bit_ids)

    def _tile(self, tile):
        if isinstance(tile, (int, np.integer)):
            return self.tile_ids[int(tile)] if 0 <= int(tile) < len(self.tile_ids) else None
        return tile if tile in self.tile_ids else None

    def _round(self, rnd):
        if isinstance(rnd, (int, np.integer)):
            return self.round_ids[int(rnd)] if 0 <= int(rnd) < len(self.round_ids) else None
        return rnd if rnd in self.round_ids else None

    def _bit(self, bit):
        if isinstance(bit, (int, np.integer)):
            return self.bit_ids[int(bit)] if 0 <= int(bit) < len(self.bit_ids) else None
        return bit if bit in self.bit_ids else None

    @staticmethod
    def _out(value, return_future):
        return _Ready(value) if return_future else value

    def _key(self, tile, round=None, bit=None):
        t = self._tile(tile)
        if t is None or (round is None) == (bit is None):
            print("Provide a valid tile and either 'round' or 'bit'")  # store convention: print + None
            return None
        if round is not None:
            r = self._round(round)
            return None if r is None else (t, "round", r)
        b = self._bit(bit)
        return None if b is None else (t, "bit", b)

    # ---- images -------------------------------------------------------------------------
    def set_corrected_image(self, image, tile, round=None, bit=None, wavelengths_um=(0.561, 0.590)):
        k = self._key(tile, round, bit)
        self._corrected[k] = image
        self._wavelengths[k] = tuple(float(v) for v in wavelengths_um)

    def save_local_corrected_image(self, image, tile, gain_correction: bool = True, hotpixel_correction: bool = True,
                                   shading_correction: bool = False, psf_idx: int = 0, round=None, bit=None,
                                   return_future=False):
        """qi2labDataStore.save_local_corrected_image (:3639-3770): the image plus its correction flags and PSF index."""
        k = self._key(tile, round, bit)
        if k is None:
            return None
        self._corrected[k] = image
        self._image_attributes(k).update({"gain_correction": bool(gain_correction),
                                          "hotpixel_correction": bool(hotpixel_correction),
                                          "shading_correction": bool(shading_correction), "psf_idx": int(psf_idx)})

    def _image_attributes(self, key) -> dict:
        if not hasattr(self, "_image_attrs"):
            self._image_attrs = {}
        return self._image_attrs.setdefault(key, {})

    def save_local_wavelengths_um(self, wavelengths_um, tile, round=None, bit=None):
        k = self._key(tile, round, bit)
        if k is not None:
            self._wavelengths[k] = tuple(float(v) for v in wavelengths_um)

    def save_local_stage_position_zyx_um(self, stage_zyx_um, affine_zyx_px, tile, round):
        self.set_stage_position_zyx_um(tile, stage_zyx_um, affine_zyx_px, round=round)

    def load_local_corrected_image(self, tile, round=None, bit=None, return_future=True):
        k = self._key(tile, round, bit)
        if k is None or k not in self._corrected:
            return None
        return self._out(self._corrected[k], return_future)

    def _image_shape(self, tile, round=None, bit=None, kind="corrected_data"):
        """Shape of a stored array or None; ``kind`` follows the on-disk folder names
        (``corrected_data``, ``decon_data``, ``<feature predictor>_data``)."""
        k = self._key(tile, round, bit)
        if k is None:
            return None
        if kind == "corrected_data":
            src = self._corrected
        elif kind in ("decon_data", "registered_decon_data"):
            src = self._readout_decon if k[1] == "bit" else self._fiducial_decon
        elif kind == f"{self.feature_predictor_folder_name}_data":
            src = self._predictor
        else:
            return None
        arr = src.get(k)
        return None if arr is None else tuple(int(v) for v in arr.shape)

    def has_feature_predictor_spots(self, tile, bit) -> bool:
        return self._key(tile, bit=bit) in self._predictor_spots

    def save_local_deconvolved_fiducial_image(self, image, tile, round, **_):
        self._fiducial_decon[self._key(tile, round=round)] = image

    def load_local_fiducial_image(self, tile, round, return_future=True):
        k = self._key(tile, round=round)
        arr = self._fiducial_decon.get(k)
        if arr is None:      # (not ``get(k, fallback)``: on a disk store the fallback would be READ before it is needed)
            arr = self._corrected.get(k)
        return None if arr is None else self._out(arr, return_future)

    def save_local_deconvolved_readout_image(self, image, tile, bit, **_):
        self._readout_decon[self._key(tile, bit=bit)] = image

    def load_local_readout_image(self, tile, bit, return_future=True):
        """Deconvolved readout if present else corrected (qi2labDataStore.py:4709)."""
        k = self._key(tile, bit=bit)
        arr = self._readout_decon.get(k)
        if arr is None:
            arr = self._corrected.get(k)
        return None if arr is None else self._out(arr, return_future)

    def save_local_feature_predictor_image(self, feature_predictor_image, tile, bit, **_):
        self._predictor[self._key(tile, bit=bit)] = feature_predictor_image

    def load_local_feature_predictor_image(self, tile, bit, return_future=True):
        arr = self._predictor.get(self._key(tile, bit=bit))
        return None if arr is None else self._out(arr, return_future)

    def save_local_feature_predictor_spots(self, spot_df, tile, bit, **_):
        self._predictor_spots[self._key(tile, bit=bit)] = spot_df

    # ---- metadata -----------------------------------------------------------------------
    def set_round_linker(self, tile, bit, round_1based: int):
        self._round_linker[(self._tile(tile), self._bit(bit))] = int(round_1based)

    def load_local_round_linker(self, tile, bit):
        return self._round_linker.get((self._tile(tile), self._bit(bit)), 1)

    def load_local_wavelengths_um(self, tile, round=None, bit=None):
        return self._wavelengths.get(self._key(tile, round, bit), (0.561, 0.590))

    def save_local_round_transform_zyx_um(self, transform_zyx_um, tile, round):
        self._round_xform[(self._tile(tile), self._round(round))] = np.asarray(transform_zyx_um, dtype=np.float32)

    def load_local_round_transform_zyx_um(self, tile, round):
        return self._round_xform.get((self._tile(tile), self._round(round)))

    def save_local_sofima_flow_field(self, sofima_flow_field_xyz_px, *, tile, round, **attrs):
        attrs.pop("return_future", None)
        self._flow[(self._tile(tile), self._round(round))] = (
            np.asarray(sofima_flow_field_xyz_px, dtype=np.float32), dict(attrs))

    def load_local_sofima_flow_field(self, tile, round, return_future=False):
        return self._flow.get((self._tile(tile), self._round(round)))

    def set_chromatic_affine_transform_zyx_um(self, wavelength_um, transform):
        self._chromatic[round(float(wavelength_um), 4)] = np.asarray(transform, dtype=np.float32)

    def load_chromatic_affine_transform_zyx_um(self, wavelength_um):
        """Identity fallback like qi2labDataStore.py:213."""
        return self._chromatic.get(round(float(wavelength_um), 4), np.eye(4, dtype=np.float32))

    def set_stage_position_zyx_um(self, tile, origin, camera_to_stage=None, round=0):
        c2s = np.eye(4, dtype=np.float32) if camera_to_stage is None else np.asarray(camera_to_stage)
        r = round if isinstance(round, (int, np.integer)) else self.round_ids.index(round)
        self._stage[(self._tile(tile), int(r))] = (np.asarray(origin, dtype=np.float32), c2s)

    def load_local_stage_position_zyx_um(self, tile, round=0):
        r = round if isinstance(round, (int, np.integer)) else self.round_ids.index(round)
        return self._stage.get((self._tile(tile), int(r)))

    def set_global_coord_xforms_um(self, tile, affine, origin, spacing):
        self._global_xform[self._tile(tile)] = (np.asarray(affine), np.asarray(origin), np.asarray(spacing))

    def load_global_coord_xforms_um(self, tile):
        return self._global_xform.get(self._tile(tile), (None, None, None))

    # ---- decode products -------------------------------------------------------------------
    def save_decode_normalization_vectors(self, decode_run_key, kind, normalization, background,
                                          decode_mode=None, metadata=None):
        self._norm_vectors[(decode_run_key, kind)] = (np.asarray(normalization).copy(),
                                                      np.asarray(background).copy())
        # the store keeps the metadata mapping as given (qi2labDataStore.py:1229-1232, 1253-1271);
        # decode_mode is a separate run attribute
        self._norm_meta[(decode_run_key, kind)] = None if metadata is None else dict(metadata)
        if decode_mode is not None:
            self._norm_decode_mode[decode_run_key] = str(decode_mode)

    def load_decode_normalization_vectors(self, decode_run_key, kind):
        return self._norm_vectors.get((decode_run_key, kind), (None, None))

    def load_decode_normalization_metadata(self, decode_run_key, kind):
        meta = self._norm_meta.get((decode_run_key, kind))
        return dict(meta) if meta is not None else None

    def decoded_temporary_dir(self, decode_run_key=None):
        return self._datastore_path / "decoded_temp" / str(decode_run_key or "default")

    def save_local_decoded_spots(self, df, tile, decode_run_key=None):
        self._decoded[(decode_run_key, self._tile(tile))] = df.copy()

    def load_local_decoded_spots(self, tile, decode_run_key=None):
        return self._decoded.get((decode_run_key, self._tile(tile)))

    def save_global_filtered_decoded_spots(self, df, decode_run_key=None):
        self._filtered[decode_run_key] = df.copy()

    def load_global_filtered_decoded_spots(self, decode_run_key=None):
        return self._filtered.get(decode_run_key)

    def load_global_cellpose_outlines(self):
        return None

    def load_global_cellpose_roi_zip(self):
        return None


class _DiskImages:
    """Dictionary face of one image family kept as ``<entity>/<name>.ome.zarr`` stores on disk."""

    def __init__(self, owner: "ZarrImageDataStore", name: str):
        self._owner, self._name = owner, name

    def path(self, key) -> Path:
        tile, kind, ident = key
        root = self._owner._fiducial_root_path if kind == "round" else self._owner._readouts_root_path
        return root / tile / ident / self._name

    def __contains__(self, key) -> bool:
        from . import zarr_io

        if key is None:
            return False
        path = self.path(key)
        if path in self._owner._resident or path in self._owner._pending_writes:
            return True
        return (zarr_io.image_store_path(path) / "0" / "zarr.json").exists()

    def __setitem__(self, key, image) -> None:
        self._owner._drop_prefetched(self.path(key))
        self._owner._write(self.path(key), image)

    def __getitem__(self, key):
        path = self.path(key)
        kept = self._owner._resident.get(path)
        if kept is not None:                       # written by this process and still in HBM: no I/O
            return kept
        ready = self._owner._take_prefetched(path)
        if ready is not None:
            return ready
        self._owner._wait_for_write(path)          # queued behind the compute: let it land first
        return self._owner._io.read_image(path, device=self._owner._device)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def shape(self, key):
        if key is None or key not in self:
            return None
        kept = self._owner._resident.get(self.path(key))
        if kept is not None:
            return tuple(int(v) for v in kept.shape)
        self._owner._wait_for_write(self.path(key))
        return self._owner._io.image_shape(self.path(key))


class ZarrImageDataStore(InMemoryDataStore):
    """The same duck type with every image on disk in the reference's layout and format
    (``<root>/fiducial/<tile>/<round>/{corrected_data,decon_data,opticalflow_xform_px}.ome.zarr``,
    ``<root>/readouts/<tile>/<bit>/{corrected_data,decon_data,<predictor>_data}.ome.zarr``;
    qi2labDataStore.py:1310-1332, 3592-3619, 4087, 4491, 4690), read and written through the
    I/O fast path (``zarr_io.ImageIO``).  ``device="cuda:0"`` makes every ``load_*`` return a tensor
    already in HBM (zstd on host threads, bit shuffle + chunk scatter on the GPU).  Scalars, transforms and
    tables stay in memory: the reference's JSON / parquet sidecars are outside the hot path.
    """

    def __init__(self, datastore_path, *, device=None, num_threads=None, write_behind: bool = False,
                 prefetch_threads: int | None = None, write_threads: int | None = None,
                 max_pending_writes: int = 6, **kwargs):
        """``write_behind=True`` (needs ``device``): image saves return at once - the volume is handed to a
        background writer with its own HIP stream and zstd threads and stays readable from HBM (``_resident``)
        until ``evict_tile`` - so a tile's I/O runs under its own and the next tile's kernels.  ``flush()`` waits for
        the queue; a load of an image whose write is still queued waits for that write only.  The reference writes
        synchronously (qi2labDataStore.py:2273-2373); results on disk are byte-identical either way."""
        from . import zarr_io

        super().__init__(datastore_path=datastore_path, **kwargs)
        self._device = device
        self._io = zarr_io.ImageIO(num_threads=num_threads)
        self._prefetched = {}        # image path -> Future[tensor in HBM]
        self._prefetch_pool = None
        self._prefetch_io = None
        self._prefetch_stream = None
        self._prefetch_threads = prefetch_threads
        self._write_behind = bool(write_behind) and device is not None
        self._write_threads = write_threads
        self._max_pending_writes = int(max_pending_writes)
        self._resident = {}          # image path -> tensor in HBM written through this store (write-behind only)
        self._pending_writes = {}    # image path -> Future of the background write
        self._write_pool = None
        self._write_io = None
        self._write_stream = None
        self.io_seconds = {"read_wait": 0.0, "write_wait": 0.0}   # *_wait: time the CALLER spent waiting on I/O;
        self.io_counts = {}                                        # *_busy: time the background threads spent in it
        self.fiducial_folder_name = "fiducial"
        self._fiducial_root_path = self._datastore_path / self.fiducial_folder_name
        self._readouts_root_path = self._datastore_path / "readouts"
        self._feature_predictor_localizations_root_path = (
            self._datastore_path / f"{self.feature_predictor_folder_name}_localizations")
        for p in (self._fiducial_root_path, self._readouts_root_path, self._feature_predictor_localizations_root_path):
            p.mkdir(parents=True, exist_ok=True)
        self._corrected = _DiskImages(self, "corrected_data")
        self._fiducial_decon = _DiskImages(self, "decon_data")
        self._readout_decon = _DiskImages(self, "decon_data")
        self._predictor = _DiskImages(self, f"{self.feature_predictor_folder_name}_data")

    # a spawned per-GPU worker receives the operator (and with it this store) by pickle
    # (DataRegistration.py:2151-2154): the reader handle, its staging and the read-ahead thread are rebuilt there
    def __getstate__(self):
        state = dict(self.__dict__)
        state["_io_threads"] = self._io.num_threads
        self.flush()
        for key in ("_io", "_prefetch_pool", "_prefetch_io", "_prefetch_stream", "_write_pool", "_write_io",
                    "_write_stream"):
            state[key] = None
        state["_prefetched"] = {}
        state["_resident"] = {}
        state["_pending_writes"] = {}
        return state

    def __setstate__(self, state):
        from . import zarr_io

        threads = state.pop("_io_threads", None)
        self.__dict__.update(state)
        self._io = zarr_io.ImageIO(num_threads=threads)

    def _image_shape(self, tile, round=None, bit=None, kind="corrected_data"):
        """By keys like the in-memory store, or - the reference's form (qi2labDataStore.py:1835-1862) -
        by the logical image path ``<entity root>/<image name>``."""
        if round is None and bit is None:
            from . import zarr_io

            kept = self._resident.get(Path(tile))
            if kept is not None:
                return tuple(int(v) for v in kept.shape)
            self._wait_for_write(Path(tile))
            path = zarr_io.image_store_path(tile)
            return self._io.image_shape(path) if (path / "0" / "zarr.json").exists() else None
        k = self._key(tile, round, bit)
        if k is None:
            return None
        if kind == "corrected_data":
            return self._corrected.shape(k)
        if kind in ("decon_data", "registered_decon_data"):
            return (self._readout_decon if k[1] == "bit" else self._fiducial_decon).shape(k)
        if kind == f"{self.feature_predictor_folder_name}_data":
            return self._predictor.shape(k)
        return None

    # ---- read-ahead: the next tile's images are loaded on a side stream while this tile computes ----
    def prefetch_tile(self, tile, bits=None) -> int:
        """Start loading the images ``decode_one_tile`` will ask for (per bit: the deconvolved readout if it
        exists, else the corrected one, and the predictor image) into HBM; returns how many loads were queued.
        The loads run in one background thread with its own reader (zstd on its host threads) and its own
        HIP stream, so PCIe copies and the unshuffle kernels overlap the current tile's kernels."""
        if self._device is None:
            return 0
        from concurrent.futures import ThreadPoolExecutor

        if self._prefetch_pool is None:
            self._prefetch_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="m3d-prefetch")
        t = self._tile(tile)
        queued = 0
        for bit in (self.bit_ids if bits is None else bits):
            k = self._key(t, bit=bit)
            if k is None:
                continue
            src = self._readout_decon if k in self._readout_decon else self._corrected
            for family in (src, self._predictor):
                path = family.path(k)
                if k in family and path not in self._prefetched:
                    self._prefetched[path] = self._prefetch_pool.submit(self._prefetch_one, path)
                    queued += 1
        return queued

    def prefetch_tile_inputs(self, tile, rounds=None, bits=None) -> int:
        """Read-ahead for the PREPROCESS of ``tile`` (DataRegistration.py:390-453, 906-1121): the corrected fiducial
        image of every round, then per bit the corrected readout and the predictor image (if one is stored), queued
        in the order the operators ask for them.  Returns the number of loads queued."""
        if self._device is None:
            return 0
        t = self._tile(tile)
        paths = []
        for rnd in (self.round_ids if rounds is None else rounds):
            k = self._key(t, round=rnd)
            if k is not None and k in self._corrected:
                paths.append(self._corrected.path(k))
        for bit in (self.bit_ids if bits is None else bits):
            k = self._key(t, bit=bit)
            if k is None:
                continue
            for family in (self._corrected, self._predictor):
                if k in family:
                    paths.append(family.path(k))
        return self.prefetch_images(paths)

    def prefetch_images(self, paths) -> int:
        from concurrent.futures import ThreadPoolExecutor

        if self._device is None:
            return 0
        if self._prefetch_pool is None:
            self._prefetch_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="m3d-prefetch")
        queued = 0
        for path in paths:
            if path not in self._prefetched and path not in self._resident:
                self._prefetched[path] = self._prefetch_pool.submit(self._prefetch_one, path)
                queued += 1
        return queued

    def _prefetch_one(self, path):
        import torch

        from . import zarr_io

        self._wait_for_write(path, account=False)
        if self._prefetch_io is None:
            self._prefetch_io = zarr_io.ImageIO(num_threads=self._prefetch_threads or self._io.num_threads)
            self._prefetch_stream = torch.cuda.Stream(device=self._device)
        import time

        t0 = time.perf_counter()
        with torch.cuda.stream(self._prefetch_stream):
            out = self._prefetch_io.read_image(path, device=self._device)   # returns after its stream drained
        st = self._prefetch_io.last_stats
        self.io_seconds["read_busy"] = self.io_seconds.get("read_busy", 0.0) + time.perf_counter() - t0
        self.io_counts["images_read"] = self.io_counts.get("images_read", 0) + 1
        self.io_counts["raw_bytes_read"] = self.io_counts.get("raw_bytes_read", 0) + int(st.get("raw_bytes", 0))
        return out

    def _take_prefetched(self, path):
        fut = self._prefetched.pop(path, None)
        if fut is None:
            return None
        import torch

        import time

        t0 = time.perf_counter()
        tensor = fut.result()
        self.io_seconds["read_wait"] += time.perf_counter() - t0
        tensor.record_stream(torch.cuda.current_stream(tensor.device))  # allocated under the side stream
        return tensor

    # ---- write-behind: saves are queued to a background writer, the volume stays readable from HBM ----
    def _write(self, path, image) -> None:
        import torch

        spacing = self.voxel_size_zyx_um
        if not (self._write_behind and isinstance(image, torch.Tensor) and image.is_cuda):
            self._wait_for_write(path)
            self._resident.pop(path, None)
            self._io.write_image(path, image, ome_scale=spacing)
            return
        import time
        from concurrent.futures import ThreadPoolExecutor

        if self._write_pool is None:
            self._write_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="m3d-writer")
        self._wait_for_write(path)                       # an older version of the same image goes first
        t0 = time.perf_counter()
        while len(self._pending_writes) >= self._max_pending_writes:   # back-pressure: bounded HBM held by the queue
            oldest = next(iter(self._pending_writes))
            self._pending_writes.pop(oldest).result()
        self.io_seconds["write_wait"] += time.perf_counter() - t0
        image = image.contiguous()
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(image.device))         # the producer's kernels, as enqueued so far
        self._resident[path] = image
        self._pending_writes[path] = self._write_pool.submit(self._write_one, path, image, ready, spacing)

    def _write_one(self, path, image, ready, spacing):
        import torch

        from . import zarr_io

        if self._write_io is None:
            self._write_io = zarr_io.ImageIO(num_threads=self._write_threads or self._io.num_threads)
            self._write_stream = torch.cuda.Stream(device=self._device)
        import time

        t0 = time.perf_counter()
        with torch.cuda.device(image.device), torch.cuda.stream(self._write_stream):
            self._write_stream.wait_event(ready)
            ready.synchronize()          # (not part of the write: the producer may still be computing this volume)
            t1 = time.perf_counter()
            self._write_io.write_image(path, image, ome_scale=spacing)   # returns when the last chunk is on disk
        st = self._write_io.last_stats
        self.io_seconds["write_busy"] = self.io_seconds.get("write_busy", 0.0) + time.perf_counter() - t1
        self.io_seconds["write_entropy"] = self.io_seconds.get("write_entropy", 0.0) + float(st.get("seconds_entropy", 0.0))
        self.io_seconds["write_producer_wait"] = self.io_seconds.get("write_producer_wait", 0.0) + t1 - t0
        self.io_counts["images_written"] = self.io_counts.get("images_written", 0) + 1
        self.io_counts["raw_bytes_written"] = self.io_counts.get("raw_bytes_written", 0) + int(st.get("raw_bytes", 0))
        self.io_counts["file_bytes_written"] = self.io_counts.get("file_bytes_written", 0) + int(st.get("file_bytes", 0))
        return path

    def _wait_for_write(self, path, account: bool = True) -> None:
        fut = self._pending_writes.get(path)
        if fut is None:
            return
        import time

        t0 = time.perf_counter()
        fut.result()
        self._pending_writes.pop(path, None)
        if account:
            self.io_seconds["write_wait"] += time.perf_counter() - t0

    def flush(self) -> None:
        """Wait until every queued image write is on disk (errors of the background writer surface here)."""
        import time

        t0 = time.perf_counter()
        for path in list(self._pending_writes):
            fut = self._pending_writes.pop(path, None)
            if fut is not None:
                fut.result()
        self.io_seconds["write_wait"] += time.perf_counter() - t0

    def evict_tile(self, tile) -> int:
        """Drop the HBM copies this store kept of ``tile``'s written images (queued writes keep their own
        reference until they are on disk) and any read-ahead that was not consumed; returns the number dropped."""
        t = self._tile(tile)
        if t is None:
            return 0
        n = 0
        for path in [p for p in self._resident if p.parent.parent.name == t]:
            del self._resident[path]
            n += 1
        for path in [p for p in self._prefetched if p.parent.parent.name == t]:
            self._drop_prefetched(path)
        return n

    def _drop_prefetched(self, path) -> None:
        fut = self._prefetched.pop(path, None)
        if fut is not None:
            fut.result()

    # ---- predictor localisations: <root>/<predictor>_localizations/<tile>/<bit>.parquet (:5072-5130) ----
    def _spots_path(self, tile, bit) -> Path:
        return self._feature_predictor_localizations_root_path / self._tile(tile) / f"{self._bit(bit)}.parquet"

    def save_local_feature_predictor_spots(self, spot_df, tile, bit, **_):
        if self._tile(tile) is None or self._bit(bit) is None:
            print("Set valid tile and bit ids")
            return None
        path = self._spots_path(tile, bit)
        path.parent.mkdir(parents=True, exist_ok=True)
        self._save_to_parquet(spot_df, path)

    @staticmethod
    def _save_to_parquet(df, path) -> None:
        df.to_parquet(path)

    def load_local_feature_predictor_spots(self, tile, bit):
        path = self._spots_path(tile, bit)
        return pd.read_parquet(path) if path.exists() else None

    def has_feature_predictor_spots(self, tile, bit) -> bool:
        return self._tile(tile) is not None and self._bit(bit) is not None and self._spots_path(tile, bit).exists()

    def _flow_path(self, tile, round) -> Path:
        return self._fiducial_root_path / self._tile(tile) / self._round(round) / "opticalflow_xform_px"

    def save_local_sofima_flow_field(self, sofima_flow_field_xyz_px, *, tile, round, **attrs):
        """Flow ``<f4 (3, z, y, x)`` with its map attributes as image-level attributes
        (DataRegistration.py:661-674, qi2labDataStore.py:4171-4190)."""
        attrs.pop("return_future", None)
        self._io.write_image(self._flow_path(tile, round), np.asarray(sofima_flow_field_xyz_px, dtype=np.float32),
                             extra_attributes=attrs)

    def load_local_sofima_flow_field(self, tile, round, return_future=False):
        from . import zarr_io

        path = self._flow_path(tile, round)
        if not (zarr_io.image_store_path(path) / "0" / "zarr.json").exists():
            return None
        return self._io.read_image(path), zarr_io.read_extra_attributes(path)
