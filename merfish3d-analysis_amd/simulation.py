"""statphysbio simulation -> fake acquisition -> datastore -> F1.

Host-side counterparts of the reference's simulation CLI, which is what its only end-to-end acceptance test drives
(``tests/test_simulation_example_pipeline.py``: SOFIMA must not reduce F1 below the affine-only run):

* ``convert_simulation``  - cli/statphysbio_simulation/convert_simulation_to_experiment.py:26-150: the flat
  ``aligned_1.tiff`` (rounds x channels x z pages) becomes one ``data_rNNNN_tileNNNN.tif`` per round plus the
  acquisition's metadata CSVs;
* ``convert_data``        - cli/statphysbio_simulation/convert_to_datastore.py:185-573: gain / offset correction,
  channel order, z sub-sampling, fiducial (channel 0) and the two readout channels of every round written into a
  datastore with PSFs, stage positions, wavelengths and bit <-> round links;
* ``synthetic_chromatic_affines_zyx_um`` - :42-118 of the same file;
* ``calculate_F1``        - cli/statphysbio_simulation/calculate_F1.py:207-276: decoded z shifted to voxel centres
  (+ dz / 2), ground truth read as (Z, X, Y) - its X and Y are swapped - and re-centred, both cropped to the imaged z
  range, then ``scoring.calculate_F1_with_radius``;
* ``make_simulation``     - the build's own generator of a dataset in that on-disk format (the published simulation
  is a download; there is no network here).

Pinning.  ``convert_simulation`` and ``convert_data`` are checked against a run of the reference's own functions
(tests/golden/make_golden_simulation.py -> tests/test_simulation_reference_run.py): the acquisition folder file by file,
every stored image, every JSON sidecar of the datastore tree.  The one stand-in under that run that is code of this
module - the TIFF codec below, used because the main interpreter has no tifffile - is itself pinned in both directions
against the real tifffile 2021.7.2 of the image's conda Python (tests/golden/make_golden_tiff.py): it reads what
tifffile writes (the RGB(A)-planar pages tifffile picks for a leading axis of 3 or 4 included) and tifffile reads what it
writes.  Still a stand-in: the PSFs come from ``gaussian_widefield_psf`` unless a ``psf_factory`` is passed (the
reference calls ``psfmodels.make_psf(model="vectorial")``; psfmodels is absent from every interpreter of the image).
"""

from __future__ import annotations

import json
import re
import shutil
import struct
from itertools import compress
from pathlib import Path
from typing import Callable, Sequence

import numpy as np
import pandas as pd

from .scoring import calculate_F1_with_radius

# ----------------------------------------------------------------------------
# metadata CSVs (utils/dataio.py:31-72, :95-105)
# ----------------------------------------------------------------------------
def read_metadatafile(fname) -> dict:
    """First line = keys, second line = values; ints, floats and booleans are converted (dataio.py:31-72)."""
    with open(fname) as f:
        lines = [line.replace("\n", "") for line in f]
    titles, vals = lines[0].split(","), lines[1].split(",")
    for i, v in enumerate(vals):
        if re.fullmatch(r"\d+", v):
            vals[i] = int(v)
        elif re.fullmatch(r"\d*.\d+", v):
            vals[i] = float(v)
        elif v.lower() == "false":
            vals[i] = False
        elif v.lower() == "true":
            vals[i] = True
    return dict(zip(titles, vals))


def write_metadata(data_dict: dict, save_path) -> None:
    """One-row CSV with pandas' index column in front, as the reference writes it (dataio.py:95-105)."""
    pd.DataFrame([data_dict]).to_csv(save_path)


# ----------------------------------------------------------------------------
# baseline TIFF (uncompressed strips; classic or BigTIFF on read, either on write)
# ----------------------------------------------------------------------------
_TIFF_DTYPES = {(1, 8): np.uint8, (1, 16): np.uint16, (1, 32): np.uint32, (2, 16): np.int16, (3, 32): np.float32}


def write_tiff(path, array: np.ndarray) -> None:
    """Pages = the last two axes of ``array``; the full shape goes into the first page's ImageDescription the way
    tifffile's ``imwrite`` records it, so either program reads the other's files."""
    a = np.ascontiguousarray(array)
    if a.dtype not in (np.uint8, np.uint16, np.float32):
        raise ValueError(f"write_tiff: unsupported dtype {a.dtype}")
    if a.ndim < 2:
        raise ValueError("write_tiff needs at least two dimensions")
    ny, nx = a.shape[-2:]
    pages = a.reshape(-1, ny, nx)
    bits = a.dtype.itemsize * 8
    fmt = 3 if a.dtype == np.float32 else 1
    desc = (json.dumps({"shape": list(a.shape)}) + "\0").encode()
    page_bytes = ny * nx * a.dtype.itemsize
    big = pages.shape[0] * (page_bytes + 512) + len(desc) > 0xF0000000
    with open(path, "wb") as f:
        if big:
            f.write(struct.pack("<2sHHHQ", b"II", 43, 8, 0, 16))
        else:
            f.write(struct.pack("<2sHI", b"II", 42, 8))
        for p in range(pages.shape[0]):
            here = f.tell()
            tags = [(256, 4, 1, nx), (257, 4, 1, ny), (258, 3, 1, bits), (259, 3, 1, 1), (262, 3, 1, 1),
                    (273, 16 if big else 4, 1, None), (277, 3, 1, 1), (278, 4, 1, ny),
                    (279, 16 if big else 4, 1, page_bytes), (339, 3, 1, fmt)]
            if p == 0:
                tags.append((270, 2, len(desc), None))
            tags.sort()
            entry = 20 if big else 12
            ifd_bytes = (8 if big else 2) + entry * len(tags) + (8 if big else 4)
            desc_at = here + ifd_bytes
            data_at = desc_at + (len(desc) if p == 0 else 0)
            data_at += (-data_at) % 16
            nxt = 0 if p == pages.shape[0] - 1 else data_at + page_bytes
            f.write(struct.pack("<Q" if big else "<H", len(tags)))
            for tag, typ, count, value in tags:
                if tag == 273:
                    value = data_at
                if tag == 270:
                    value = desc_at
                if big:
                    f.write(struct.pack("<HHQQ", tag, typ, count, value))
                else:
                    f.write(struct.pack("<HHI", tag, typ, count))
                    f.write(struct.pack("<HH", value, 0) if typ == 3 else struct.pack("<I", value))
            f.write(struct.pack("<Q" if big else "<I", nxt))
            if p == 0:
                f.write(desc)
            f.write(b"\0" * (data_at - f.tell()))
            f.write(pages[p].tobytes())


def read_tiff(path) -> np.ndarray:
    """Uncompressed little- or big-endian TIFF / BigTIFF; pages are stacked and reshaped by the ``{"shape": ...}``
    description when there is one."""
    with open(path, "rb") as f:
        buf = f.read()
    order = {b"II": "<", b"MM": ">"}.get(buf[:2])
    if order is None:
        raise ValueError(f"{path}: not a TIFF file")
    magic = struct.unpack(order + "H", buf[2:4])[0]
    big = magic == 43
    off = struct.unpack(order + "Q", buf[8:16])[0] if big else struct.unpack(order + "I", buf[4:8])[0]
    sizes = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 16: 8, 11: 4, 12: 8}
    codes = {1: "B", 3: "H", 4: "I", 16: "Q", 11: "f", 12: "d"}
    pages, shape_hint = [], None
    while off:
        n = struct.unpack(order + ("Q" if big else "H"), buf[off:off + (8 if big else 2)])[0]
        pos = off + (8 if big else 2)
        tags = {}
        for _ in range(n):
            if big:
                tag, typ, count = struct.unpack(order + "HHQ", buf[pos:pos + 12])
                field, inline, step = buf[pos + 12:pos + 20], 8, 20
            else:
                tag, typ, count = struct.unpack(order + "HHI", buf[pos:pos + 8])
                field, inline, step = buf[pos + 8:pos + 12], 4, 12
            nbytes = sizes.get(typ, 1) * count
            raw = field[:nbytes] if nbytes <= inline else None
            if raw is None:
                at = struct.unpack(order + ("Q" if big else "I"), field)[0]
                raw = buf[at:at + nbytes]
            if typ == 2:
                tags[tag] = raw.split(b"\0")[0].decode(errors="replace")
            elif typ in codes:
                vals = struct.unpack(order + codes[typ] * count, raw)
                tags[tag] = vals[0] if count == 1 else vals
            pos += step
        off = struct.unpack(order + ("Q" if big else "I"), buf[pos:pos + (8 if big else 4)])[0]
        if tags.get(259, 1) != 1:
            raise ValueError(f"{path}: compressed TIFF pages are not supported")
        nx, ny = int(tags[256]), int(tags[257])
        first = lambda v, d: d if v is None else (v[0] if isinstance(v, tuple) else v)  # noqa: E731
        samples = int(tags.get(277, 1))
        bits, fmt = tags.get(258, 1), tags.get(339, 1)
        if any(isinstance(v, tuple) and len(set(v)) != 1 for v in (bits, fmt)):
            raise ValueError(f"{path}: samples of different types in one page")
        dtype = _TIFF_DTYPES.get((int(first(fmt, 1)), int(first(bits, 1))))
        if dtype is None:
            raise ValueError(f"{path}: unsupported sample layout")
        offs, cnts = tags[273], tags.get(279, samples * ny * nx * np.dtype(dtype).itemsize)
        offs = offs if isinstance(offs, tuple) else (offs,)
        cnts = cnts if isinstance(cnts, tuple) else (cnts,)
        raw = b"".join(buf[o:o + c] for o, c in zip(offs, cnts))
        page = np.frombuffer(raw, dtype=np.dtype(dtype).newbyteorder(order), count=samples * ny * nx)
        # tifffile stores a leading axis of length 3 or 4 as the samples of an RGB(A) page: separate planes
        # (PlanarConfiguration 2, what it chooses for (3, y, x) stacks) or interleaved
        if samples == 1:
            pages.append(page.reshape(ny, nx))
        elif int(tags.get(284, 1)) == 2:
            pages.append(page.reshape(samples, ny, nx))
        else:
            pages.append(page.reshape(ny, nx, samples))
        if shape_hint is None and 270 in tags and tags[270].startswith("{"):
            try:
                shape_hint = tuple(int(v) for v in json.loads(tags[270])["shape"])
            except (ValueError, KeyError, TypeError):
                shape_hint = None
    out = np.stack(pages).astype(pages[0].dtype.newbyteorder("="), copy=False)
    if shape_hint is not None and int(np.prod(shape_hint)) == out.size:
        out = out.reshape(shape_hint)
    return out


# ----------------------------------------------------------------------------
# simulation -> fake acquisition (convert_simulation_to_experiment.py:26-150)
# ----------------------------------------------------------------------------
def convert_simulation(root_path) -> Path:
    """Write ``root_path/sim_acquisition`` from ``aligned_1.tiff`` + ``scan_metadata.csv`` and return its path."""
    root_path = Path(root_path)
    md = read_metadatafile(root_path / "scan_metadata.csv")
    root_name, num_rounds, num_ch, num_z = md["root_name"], md["num_r"], md["num_ch"], md["planes per bit"]
    yx_pixel_um, z_pixel_um = md["pixel_size [micron]"], md["axial_step_size [micron]"]
    gain, offset = md["mean gain"], md["mean offset"]
    data = read_tiff(root_path / "aligned_1.tiff")
    data = data.reshape(num_rounds, num_ch, num_z, data.shape[-2], data.shape[-1])
    stage_zyx = [0.0, -1 * yx_pixel_um * (data.shape[-2] // 2), -1 * yx_pixel_um * (data.shape[-1] // 2)]
    tile = 0
    acq = root_path / "sim_acquisition"
    acq.mkdir(exist_ok=True)
    for r in range(num_rounds):
        stem = "data_r" + str(r + 1).zfill(4) + "_tile" + str(tile).zfill(4)
        tile_path = acq / (stem + "_1")
        tile_path.mkdir(exist_ok=True)
        write_tiff(tile_path / (stem + ".tif"), np.squeeze(data[r]).astype(np.uint16))
        write_metadata({"stage_x": float(stage_zyx[2]), "stage_y": float(stage_zyx[1]), "stage_z": float(stage_zyx[0]),
                        "offset_z": 0.0, "blue_active": True, "yellow_active": True, "red_active": True},
                       acq / (stem + "_stage_positions.csv"))
    write_metadata({
        "root_name": str(root_name), "scan_type": "synthetic",
        "experiment_type": "3D" if float(z_pixel_um) < 0.5 else "2D", "camera": "simulated",
        "channels_reversed": True, "stage_flipped_x": False, "stage_flipped_y": False, "image_rotated": False,
        "image_flipped_y": False, "image_flipped_x": False, "num_t": 1, "num_r": int(num_rounds), "num_xyz": 1,
        "num_z": int(num_z), "num_ch": int(num_ch), "na": 1.35, "ri": 1.51, "z_step_um": float(z_pixel_um),
        "yx_pixel_um": float(yx_pixel_um), "binning": 1, "gain": float(gain), "offset": float(offset), "overlap": 0.2,
        "blue_active": True, "yellow_active": True, "red_active": True, "blue_exposure": 200, "yellow_exposure": 200,
        "red_exposure": 200}, acq / "scan_metadata.csv")
    shutil.copy(root_path / "codebook.csv", acq / "codebook.csv")
    shutil.copy(root_path / "bit_order.csv", acq / "bit_order.csv")
    return acq


# ----------------------------------------------------------------------------
# acquisition -> datastore (convert_to_datastore.py)
# ----------------------------------------------------------------------------
def synthetic_chromatic_affines_zyx_um(image_shape_zyx, voxel_size_zyx_um, emission_wavelengths_um, shift_scale=1.0,
                                       edge_shifts_px=(0.0, 0.5, 1.0), z_shifts_px=(0.0, 0.25, 0.5),
                                       max_z_shifts_px=(0.0, 1.25, 2.5)) -> dict:
    """Channel-to-reference affines of the deterministic synthetic aberration (convert_to_datastore.py:42-118): radial
    XY scaling about the image centre plus a capped Z translation per sorted wavelength; identity for the shortest."""
    spacing = np.asarray(voxel_size_zyx_um, dtype=np.float32)
    shape = np.asarray(image_shape_zyx, dtype=np.float32)
    center_um = (shape - 1.0) * spacing / 2.0
    yx_radius_px = max(float(shape[1] - 1.0), float(shape[2] - 1.0)) / 2.0
    affines = {}
    wavelengths = sorted(float(v) for v in emission_wavelengths_um)
    for i, wavelength in enumerate(wavelengths):
        k = min(i, len(edge_shifts_px) - 1)
        edge_shift_px = edge_shifts_px[k] * float(shift_scale)
        z_shift_px = min(z_shifts_px[k] * float(shift_scale), max_z_shifts_px[k])
        if np.isclose(wavelength, wavelengths[0]):
            edge_shift_px = z_shift_px = 0.0
        correction_scale = 1.0 / (1.0 + edge_shift_px / yx_radius_px)
        affine = np.eye(4, dtype=np.float32)
        affine[1, 1] = affine[2, 2] = correction_scale
        affine[1, 3] = center_um[1] * (1.0 - correction_scale)
        affine[2, 3] = center_um[2] * (1.0 - correction_scale)
        affine[0, 3] = -z_shift_px * spacing[0]
        affines[wavelength] = affine
    return affines


def gaussian_widefield_psf(z: int, nx: int, dxy: float, dz: float, NA: float, wvl: float, ni: float) -> np.ndarray:
    """Gaussian stand-in for ``psfmodels.make_psf(model="vectorial")`` (absent here): sigma_xy = 0.21 lambda / NA,
    sigma_z = 0.66 lambda n / NA^2 (the usual paraxial fit), unit sum, shape (z, nx, nx)."""
    sxy, sz = 0.21 * wvl / NA / dxy, 0.66 * wvl * ni / (NA * NA) / dz
    zz, yy, xx = np.meshgrid(np.arange(z, dtype=np.float64) - z // 2, np.arange(nx, dtype=np.float64) - nx // 2,
                             np.arange(nx, dtype=np.float64) - nx // 2, indexing="ij")
    psf = np.exp(-0.5 * ((zz / sz) ** 2 + (yy / sxy) ** 2 + (xx / sxy) ** 2))
    return (psf / psf.sum()).astype(np.float32)


def convert_data(root_path, datastore_factory: Callable | None = None, *, codebook_path=None, bit_order_path=None,
                 z_step: int = 1, synthetic_chromatic_aberration: bool = False,
                 synthetic_chromatic_aberration_scale: float = 1.0,
                 synthetic_chromatic_edge_shifts_px=(0.0, 0.5, 1.0), synthetic_chromatic_z_shifts_px=(0.0, 0.25, 0.5),
                 synthetic_chromatic_max_z_shifts_px=(0.0, 1.25, 2.5),
                 excitation_wavelengths_um=(0.488, 0.561, 0.635), emission_wavelengths_um=(0.520, 0.580, 0.670),
                 three_dimensional_z_spacing_threshold_um: float = 0.5, psf_factory: Callable | None = None,
                 channel_names: Sequence[str] | None = None, default_tile_overlap: float = 0.2):
    """Acquisition folder -> datastore (convert_to_datastore.py:185-573) and return the datastore.

    ``datastore_factory`` builds the store (default: ``datastore.InMemoryDataStore``; ``ZarrImageDataStore`` /
    ``Qi2labDiskStore`` take the same keywords after their path).  It is called with the keywords
    ``num_tiles, num_rounds, num_bits, codebook, voxel_size_zyx_um, microscope_type`` and - the experiment constants the
    reference writes at creation (convert_to_datastore.py:271-330) - ``experiment_order, channels_in_data,
    camera_model, tile_overlap, binning, na, ri, e_per_ADU``; the store it returns must answer
    ``save_local_corrected_image``, ``save_local_wavelengths_um``, ``save_local_stage_position_zyx_um`` and derive every
    bit's round link from ``experiment_order`` (``initialize_tile`` in the reference).  A store that exposes
    ``set_round_linker`` but has no link for a bit after creation gets it set here; a bit without a link afterwards is an
    error, not a silent default.  Per round: channel 0 is the fiducial image (PSF 0), channels 1 and 2 the readouts of
    the bits ``bit_order[round, 1:3]`` (1-based)."""
    if z_step <= 0:
        raise ValueError("z_step must be greater than 0.")
    from .datastore import InMemoryDataStore

    root_path = Path(root_path)
    codebook = pd.read_csv(codebook_path or root_path / "codebook.csv")
    experiment_order = pd.read_csv(bit_order_path or root_path / "bit_order.csv").values
    md = read_metadatafile(root_path / "scan_metadata.csv")
    root_name, num_rounds, num_tiles, num_ch, num_z = md["root_name"], md["num_r"], md["num_xyz"], md["num_ch"], md["num_z"]
    output_z_planes = len(range(0, int(num_z), int(z_step)))
    if output_z_planes <= 0:
        raise ValueError(f"z_step={z_step} does not retain any planes from num_z={num_z}.")
    e_per_adu, offset = md["gain"], md["offset"]
    channels_active = [md["blue_active"], md["yellow_active"], md["red_active"]]
    reversed_order = bool(md["channels_reversed"])
    voxel = [md["z_step_um"] * z_step, md["yx_pixel_um"], md["yx_pixel_um"]]
    na, ri = md["na"], md["ri"]
    ex, em = list(excitation_wavelengths_um), list(emission_wavelengths_um)
    channels_in_data = list(compress(range(num_ch), channels_active))
    try:
        microscope_type = md["experiment_type"]
    except KeyError:
        microscope_type = "3D" if voxel[0] < three_dimensional_z_spacing_threshold_um else "2D"
    num_bits = int(np.max(experiment_order[:, 1:]))
    try:
        tile_overlap = md["tile_overlap"]
    except KeyError:
        tile_overlap = default_tile_overlap
    factory = datastore_factory or InMemoryDataStore
    store = factory(num_tiles=int(num_tiles), num_rounds=int(num_rounds), num_bits=num_bits, codebook=codebook,
                    voxel_size_zyx_um=tuple(float(v) for v in voxel), microscope_type=str(microscope_type),
                    experiment_order=experiment_order, channels_in_data=list(channel_names or
                                                                              ["alexa488", "atto565", "alexa647"]),
                    camera_model="synthetic", tile_overlap=tile_overlap, binning=1, na=na, ri=ri, e_per_ADU=e_per_adu)
    make_psf = psf_factory or gaussian_widefield_psf
    psfs = []
    for ch in channels_in_data:
        psf = np.asarray(make_psf(z=int(output_z_planes), nx=51, dxy=voxel[1], dz=voxel[0], NA=na, wvl=em[ch], ni=ri),
                         dtype=np.float32)
        psfs.append(psf / np.sum(psf, axis=(0, 1, 2)))
    store.channel_psfs = psfs
    state = dict(store.datastore_state or {})
    state.update({"Calibrations": True})
    store.datastore_state = state
    identity_px = np.eye(4, dtype=np.float32)
    raw_shape = None
    for r in range(num_rounds):
        positions = []
        for t in range(num_tiles):
            sp = read_metadatafile(root_path / (root_name + "_r" + str(r + 1).zfill(4) + "_tile" + str(t).zfill(4)
                                                + "_stage_positions.csv"))
            stage_x, stage_y = np.round(float(sp["stage_x"]), 2), np.round(float(sp["stage_y"]), 2)
            positions.append([stage_y, stage_y, stage_x])  # the reference reads stage_y for z as well (:383)
        positions = np.asarray(positions)
        for t in range(num_tiles):
            stem = "data_r" + str(r + 1).zfill(4) + "_tile" + str(t).zfill(4)
            raw = read_tiff(root_path / (root_name + "_r" + str(r + 1).zfill(4) + "_tile" + str(t).zfill(4) + "_1")
                            / (stem + ".tif"))
            if raw_shape is None:
                raw_shape = raw.shape
            elif raw.shape != raw_shape:
                if raw.shape[0] < raw_shape[0]:
                    print(f"\nround={r + 1}; tile={t + 1}\nFound shape: {raw.shape}\nCorrect shape: {raw_shape}\n"
                          "Replacing data with zeros.\n")
                    raw = np.zeros(raw_shape, dtype=np.uint16)
                else:
                    raw = raw[:, raw.shape[1] - raw_shape[1]:, :].copy()
            if reversed_order:
                raw = np.flip(raw, axis=0)
            raw = raw[:, ::z_step, :, :]
            raw = (raw.astype(np.float32) - offset) * e_per_adu
            raw[raw < 0.0] = 0.0
            raw = raw.astype(np.uint16)
            stage = np.asarray([np.round(positions[t, 0], 2), np.round(positions[t, 1], 2),
                                np.round(positions[t, 2], 2)], dtype=np.float32)
            tile_id, round_id = store.tile_ids[t], store.round_ids[r]
            store.save_local_corrected_image(np.squeeze(raw[0]).astype(np.uint16), tile=tile_id, psf_idx=0,
                                             gain_correction=True, hotpixel_correction=False, shading_correction=False,
                                             round=round_id)
            store.save_local_stage_position_zyx_um(stage, identity_px, tile=tile_id, round=round_id)
            store.save_local_wavelengths_um((ex[0], em[0]), tile=tile_id, round=round_id)
            for ch in (1, 2):
                image = np.squeeze(raw[ch]).astype(np.uint16)
                if synthetic_chromatic_aberration:
                    image = _apply_synthetic_chromatic_aberration(
                        image, em[ch], voxel, em[1:], synthetic_chromatic_aberration_scale,
                        synthetic_chromatic_edge_shifts_px, synthetic_chromatic_z_shifts_px,
                        synthetic_chromatic_max_z_shifts_px)
                bit = int(experiment_order[r, ch]) - 1
                bit_id = store.bit_ids[bit]
                store.save_local_corrected_image(image, tile=tile_id, psf_idx=ch, gain_correction=True,
                                                 hotpixel_correction=False, shading_correction=False, bit=bit_id)
                store.save_local_wavelengths_um((ex[ch], em[ch]), tile=tile_id, bit=bit_id)
                # the bit's round link comes from the experiment order the store was created with (initialize_tile in the
                # reference); a factory written to the older six-keyword contract may not have derived it
                want_round = int(experiment_order[r, 0])
                linked = _stored_round_link(store, tile_id, bit_id)
                if linked != want_round and callable(getattr(store, "set_round_linker", None)):
                    store.set_round_linker(tile_id, bit_id, want_round)
                    linked = _stored_round_link(store, tile_id, bit_id)
                if linked != want_round:
                    raise RuntimeError(f"datastore holds round link {linked!r} for {tile_id}/{bit_id}, the bit order says "
                                       f"{want_round}: the factory must take experiment_order or expose set_round_linker")
    state = dict(store.datastore_state or {})
    state.update({"Corrected": True})
    store.datastore_state = state
    return store


def _stored_round_link(store, tile_id, bit_id):
    """The 1-based round a bit is linked to as the store itself holds it (None: no link), without the in-memory
    store's default of 1 for unknown bits."""
    from .datastore import InMemoryDataStore

    if getattr(type(store), "load_local_round_linker", None) is InMemoryDataStore.load_local_round_linker:
        return store._round_linker.get((tile_id, bit_id))
    try:
        value = store.load_local_round_linker(tile_id, bit_id)
    except (KeyError, FileNotFoundError, AttributeError):
        return None
    return None if value is None else int(value)


def _apply_synthetic_chromatic_aberration(image_zyx, emission_wavelength_um, voxel, emission_wavelengths_um, shift_scale,
                                          edge_shifts_px, z_shifts_px, max_z_shifts_px):
    """:121-182 - resample a readout image with its channel's synthetic affine, round, clip to the dtype."""
    from .utils.multiview_registration import warp_array_to_reference_gpu

    affine = synthetic_chromatic_affines_zyx_um(tuple(int(v) for v in image_zyx.shape), voxel, emission_wavelengths_um,
                                                shift_scale=shift_scale, edge_shifts_px=edge_shifts_px,
                                                z_shifts_px=z_shifts_px, max_z_shifts_px=max_z_shifts_px
                                                )[float(emission_wavelength_um)]
    if np.allclose(affine, np.eye(4, dtype=np.float32)):
        return image_zyx
    warped = warp_array_to_reference_gpu(image_zyx.astype(np.float32, copy=False), transform_zyx_um=affine,
                                         spacing_zyx_um=voxel, reference_shape=image_zyx.shape, mode="constant", cval=0.0,
                                         order=1)
    warped = np.clip(np.rint(np.asarray(warped)), 0, np.iinfo(image_zyx.dtype).max)
    return warped.astype(image_zyx.dtype, copy=False)


# ----------------------------------------------------------------------------
# F1 against the simulation's ground truth (calculate_F1.py:207-276)
# ----------------------------------------------------------------------------
def calculate_F1(datastore, gt_spots: pd.DataFrame, search_radius: float = 1.0, decoded_spots: pd.DataFrame | None = None
                 ) -> dict:
    """``gt_spots``: columns ``Z``, ``X``, ``Y`` (microns, X and Y swapped, centred on the image) and ``Gene_label``
    (1-based codebook row).  ``decoded_spots`` defaults to the datastore's filtered table."""
    if decoded_spots is None:
        decoded_spots = datastore.load_global_filtered_decoded_spots()
    gene_ids = np.asarray(datastore.codebook.iloc[:, 0].tolist())
    image = datastore.load_local_corrected_image(tile=0, round=0, return_future=False)
    shape = tuple(int(v) for v in tuple(image.shape)[-3:])
    voxel = datastore.voxel_size_zyx_um
    coords = decoded_spots[["global_z", "global_y", "global_x"]].to_numpy(dtype=np.float64).copy()
    coords[:, 0] += voxel[0] / 2
    genes = decoded_spots["gene_id"].to_numpy()
    gt = gt_spots[["Z", "X", "Y"]].to_numpy(dtype=np.float64)  # simulation GT is swapped in X & Y
    gt = gt + [0, (1.0 * shape[-2] / 2) * voxel[1] - voxel[1] / 2, (1.0 * shape[-1] / 2) * voxel[2] - voxel[2] / 2]
    gt_genes = gene_ids[gt_spots["Gene_label"].to_numpy(dtype=int) - 1]
    z_min, z_max = voxel[0] / 2, (shape[0] - 0.5) * voxel[0]
    keep = (coords[:, 0] >= z_min) & (coords[:, 0] <= z_max)
    gt_keep = (gt[:, 0] >= z_min) & (gt[:, 0] <= z_max)
    results, _, _, _ = calculate_F1_with_radius(coords[keep], genes[keep], gt[gt_keep], gt_genes[gt_keep], search_radius)
    return results


# ----------------------------------------------------------------------------
# the build's own dataset in the simulation's on-disk format
# ----------------------------------------------------------------------------
def make_simulation(root_path, *, shape_zyx=(16, 128, 128), n_bits: int = 16, n_rna: int = 300, n_beads: int = 120,
                    voxel_zyx_um=(0.315, 0.098, 0.098), round_shifts_px: Sequence[Sequence[float]] | None = None,
                    deformation_px: float = 0.0, gain: float = 1.0, offset: float = 100.0, seed: int = 0) -> Path:
    """Write ``aligned_1.tiff`` (pages: round, channel (red, yellow, blue), z), ``scan_metadata.csv``, ``codebook.csv``,
    ``bit_order.csv`` and ``GT_spots.csv`` under ``root_path``.  Blue = fiducial beads; yellow / red = the two readout
    bits of the round.  ``round_shifts_px`` displaces a round's three channels together (what registration has to
    undo); ``deformation_px`` adds a smooth lateral warp of that amplitude to the moving rounds (what SOFIMA is for)."""
    from scipy.ndimage import gaussian_filter, map_coordinates

    root_path = Path(root_path)
    root_path.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(seed)
    import itertools

    combos = list(itertools.combinations(range(n_bits), 4))
    if n_bits == 16:  # the 140 blocks of the Steiner system on GF(2)^4: weight 4, Hamming distance 4
        combos = [c for c in combos if c[0] ^ c[1] ^ c[2] ^ c[3] == 0]
    combos = combos[:140]
    mat = np.zeros((len(combos), n_bits), dtype=np.int64)
    for i, c in enumerate(combos):
        mat[i, list(c)] = 1
    n_blank = 10
    genes = [f"gene{i:03d}" for i in range(len(combos) - n_blank)] + [f"Blank-{i:02d}" for i in range(n_blank)]
    cb = pd.DataFrame(mat, columns=[f"bit{i + 1:02d}" for i in range(n_bits)])
    cb.insert(0, "gene_id", genes)
    cb.to_csv(root_path / "codebook.csv", index=False)
    n_rounds = n_bits // 2
    order = np.asarray([[r + 1, 2 * r + 1, 2 * r + 2] for r in range(n_rounds)])
    pd.DataFrame(order, columns=["round", "yellow", "red"]).to_csv(root_path / "bit_order.csv", index=False)

    Z, Y, X = shape_zyx
    dz, dy, dx = voxel_zyx_um
    code = rng.integers(0, len(combos) - n_blank, n_rna)
    pos = np.stack([rng.uniform(1.5, Z - 2.5, n_rna), rng.uniform(6, Y - 7, n_rna), rng.uniform(6, X - 7, n_rna)], axis=1)
    amp = rng.uniform(350, 900, n_rna)
    beads = np.stack([rng.uniform(1, Z - 2, n_beads), rng.uniform(4, Y - 5, n_beads), rng.uniform(4, X - 5, n_beads)], 1)

    def render(points, amps, sigma):
        vol = np.zeros(shape_zyx)
        for (pz, py, px), a in zip(points, amps):   # trilinear splat: sub-voxel positions survive
            z0, y0, x0 = int(np.floor(pz)), int(np.floor(py)), int(np.floor(px))
            fz, fy, fx = pz - z0, py - y0, px - x0
            for oz, wz in ((0, 1 - fz), (1, fz)):
                for oy, wy in ((0, 1 - fy), (1, fy)):
                    for ox, wx in ((0, 1 - fx), (1, fx)):
                        if 0 <= z0 + oz < Z and 0 <= y0 + oy < Y and 0 <= x0 + ox < X:
                            vol[z0 + oz, y0 + oy, x0 + ox] += a * wz * wy * wx
        return gaussian_filter(vol, sigma) * (2 * np.pi) ** 1.5 * np.prod(sigma)

    shifts = np.zeros((n_rounds, 3)) if round_shifts_px is None else np.asarray(round_shifts_px, dtype=float)
    zz, yy, xx = np.indices(shape_zyx, dtype=np.float64)
    pages = np.zeros((n_rounds, 3, Z, Y, X), dtype=np.uint16)
    sig = (1.2, 1.5, 1.5)
    for r in range(n_rounds):
        chans = {2: render(beads, np.full(n_beads, 2500.0), sig)}          # blue (file order: red, yellow, blue)
        for ch, bit in ((1, 2 * r), (0, 2 * r + 1)):                        # yellow = first bit, red = second
            on = mat[code, bit] > 0
            chans[ch] = render(pos[on], amp[on] * rng.uniform(0.85, 1.15, int(on.sum())), sig)
        if r > 0 and (np.any(shifts[r]) or deformation_px):
            wy = deformation_px * np.sin(2 * np.pi * xx / X + r) if deformation_px else 0.0
            wx = deformation_px * np.cos(2 * np.pi * yy / Y - r) if deformation_px else 0.0
            src = [zz - shifts[r, 0], yy - shifts[r, 1] - wy, xx - shifts[r, 2] - wx]
            chans = {k: map_coordinates(v, src, order=1, mode="nearest") for k, v in chans.items()}
        for k, v in chans.items():
            pages[r, k] = np.clip(rng.poisson(v + 8.0) / gain + offset, 0, 65535).astype(np.uint16)
    write_tiff(root_path / "aligned_1.tiff", pages.reshape(-1, Y, X))
    write_metadata({"root_name": "data", "num_r": n_rounds, "num_ch": 3, "planes per bit": Z,
                    "pixel_size [micron]": float(dy), "axial_step_size [micron]": float(dz), "mean gain": float(gain),
                    "mean offset": float(offset)}, root_path / "scan_metadata.csv")
    # ground truth as the simulation ships it: microns, centred laterally, columns named with X and Y swapped, z sampled
    # inside the voxel (voxel index z covers [z dz, (z + 1) dz): calculate_F1 adds dz / 2 to the decoded side)
    gt = pd.DataFrame({"Z": (pos[:, 0] + 0.5) * dz, "X": (pos[:, 1] - Y / 2 + 0.5) * dy, "Y": (pos[:, 2] - X / 2 + 0.5) * dx,
                       "Gene_label": code + 1})
    gt.to_csv(root_path / "GT_spots.csv", index=False)
    return root_path
