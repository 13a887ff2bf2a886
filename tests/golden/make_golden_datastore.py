"""Golden sidecar metadata written and read back by the reference's own ``qi2labDataStore`` class.

``python tests/golden/make_golden_datastore.py`` imports ``/root/reference/src/merfish3danalysis/qi2labDataStore.py``
(it imports without tensorstore / yaozarrs as long as no image is touched), builds a two-tile store through the
class's public setters and ``save_*`` methods - JSON sidecars and parquet tables only - and records

* ``files``: every JSON file of the resulting tree (relative path -> parsed content);
* ``loads``: what the reference's ``load_*`` methods and properties return for that tree;
* ``tables``: the parquet tables it wrote, as records.

``tests/test_datastore_sidecars.py`` opens the same tree with this package's ``ZarrImageDataStore`` (and writes a
second tree with it) and compares both directions.  Output: ``tests/golden/qi2lab_sidecars.json``.
"""

import json
import os
import shutil
import sys
import tempfile
from pathlib import Path

import numpy as np
import pandas as pd

sys.path.insert(0, "/root/reference/src")
from merfish3danalysis.qi2labDataStore import qi2labDataStore  # noqa: E402

here = Path(__file__).resolve().parent
# fastparquet (the engine the reference names) is not installed here: pyarrow writes the same format
qi2labDataStore._save_to_parquet = staticmethod(lambda df, path: df.to_parquet(path, engine="pyarrow", index=False))


def jsonable(v):
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    if isinstance(v, (list, tuple)):
        return [jsonable(x) for x in v]
    if isinstance(v, dict):
        return {str(k): jsonable(x) for k, x in v.items()}
    if isinstance(v, pd.DataFrame):
        return {"columns": list(v.columns), "records": jsonable(v.to_numpy().tolist())}
    if isinstance(v, Path):
        return str(v)
    return v


def main():
    root = Path(tempfile.mkdtemp(prefix="qi2lab_golden_")) / "store"
    ds = qi2labDataStore(root)
    ds.num_rounds, ds.num_tiles, ds.num_bits = 2, 2, 4
    ds.channels_in_data = ["alexa488", "atto565", "alexa647"]
    ds.microscope_type, ds.camera_model = "3D", "flir"
    ds.tile_overlap, ds.binning, ds.na, ds.ri, ds.e_per_ADU = 0.2, 1, 1.35, 1.51, 0.54
    ds.voxel_size_zyx_um = [0.315, 0.098, 0.098]
    ds.codebook = pd.DataFrame({"gene_id": ["A", "B", "Blank-1"], "bit01": [1, 0, 1], "bit02": [1, 1, 0],
                                "bit03": [0, 1, 1], "bit04": [0, 0, 0]})
    ds.experiment_order = np.asarray([[1, 1, 2], [2, 3, 4]])   # per round: fiducial round, bit (channel 2), bit (channel 3)
    for t in range(2):
        ds.initialize_tile(t)
    rng = np.random.default_rng(0)
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = [0.31, -1.2, 2.5]
    ds.save_local_round_transform_zyx_um(T, tile=1, round=1)
    ds.save_local_round_transform_zyx_um(np.eye(4, dtype=np.float32), tile=1, round=0)
    c2s = np.eye(4, dtype=np.float32)
    c2s[1, 1] = -1.0
    for r in range(2):
        ds.save_local_stage_position_zyx_um(np.asarray([10.0, 200.5 + r, -33.25], dtype=np.float32), c2s, tile=1, round=r)
        ds.save_local_wavelengths_um((0.488, 0.520), tile=1, round=r)
    for b, ex in enumerate((0.561, 0.635, 0.561, 0.635)):
        ds.save_local_wavelengths_um((ex, ex + 0.03), tile=1, bit=b)
    ds.save_local_round_linker(2, tile=1, bit=3)
    ds.save_global_coord_xforms_um(affine_zyx_um=np.eye(4, dtype=np.float32), origin_zyx_um=np.asarray([1.0, 2.0, 3.0]),
                                   spacing_zyx_um=np.asarray([0.315, 0.098, 0.098]), tile=1)
    ds.save_chromatic_affine_transforms_zyx_um({"reference_channel": "atto565", "channels": {
        "atto565": {"channel_index": 1, "wavelength_um": 0.59, "affine_zyx_um": np.eye(4).tolist()},
        "alexa647": {"channel_index": 2, "wavelength_um": 0.665, "affine_zyx_um": T.tolist()}}})
    nrm = rng.uniform(100, 300, size=4).astype(np.float32)
    bkd = rng.uniform(1, 20, size=4).astype(np.float32)
    ds.save_decode_normalization_vectors(None, "global", nrm, bkd, decode_mode="3d")
    ds.save_decode_normalization_vectors("runA", "iterative", nrm * 2, bkd * 2, decode_mode="3d",
                                         metadata={"codebook_fingerprint": "abc", "n_iterations": 3})
    spots = pd.DataFrame({"gene_id": ["A", "B"], "global_z": [1.0, 2.0], "global_y": [3.5, 4.5], "global_x": [5.0, 6.0],
                          "tile_idx": [1, 1], "distance_min": np.asarray([0.1, 0.2], dtype=np.float32)})
    ds.save_local_decoded_spots(spots, tile=1)
    ds.save_local_decoded_spots(spots.iloc[:1], tile=0, decode_run_key="runA")
    ds.save_global_filtered_decoded_spots(spots)
    ds.save_local_feature_predictor_spots(pd.DataFrame({"z": [1], "y": [2], "x": [3]}), tile=1, bit=2)
    state = ds.datastore_state
    state.update({"Calibrations": True, "Corrected": True})
    ds.datastore_state = state

    files, tables = {}, {}
    for dirpath, _dirs, names in os.walk(root):
        for name in names:
            p = Path(dirpath) / name
            rel = str(p.relative_to(root))
            if name.endswith(".json"):
                files[rel] = json.loads(p.read_text())
            elif name.endswith(".parquet"):
                tables[rel] = jsonable(pd.read_parquet(p))
    re = qi2labDataStore(root, validate=False)      # re-open: what a reader sees
    loads = {
        "tile_ids": re.tile_ids, "round_ids": re.round_ids, "bit_ids": re.bit_ids, "num_bits": re.num_bits,
        "num_rounds": re.num_rounds, "num_tiles": re.num_tiles, "microscope_type": re.microscope_type,
        "voxel_size_zyx_um": re.voxel_size_zyx_um, "codebook": re.codebook, "datastore_state": re.datastore_state,
        "round_linker": {f"1/{b}": re.load_local_round_linker(tile=1, bit=b) for b in range(4)},
        "bit_linker": {f"1/{r}": re.load_local_bit_linker(tile=1, round=r) for r in range(2)},
        "wavelengths_bit": {f"1/{b}": re.load_local_wavelengths_um(tile=1, bit=b) for b in range(4)},
        "wavelengths_round": {f"1/{r}": re.load_local_wavelengths_um(tile=1, round=r) for r in range(2)},
        "round_transform": {f"1/{r}": re.load_local_round_transform_zyx_um(tile=1, round=r) for r in range(2)},
        "round_transform_missing": re.load_local_round_transform_zyx_um(tile=0, round=1),
        "stage": {f"1/{r}": re.load_local_stage_position_zyx_um(tile=1, round=r) for r in range(2)},
        "global_xforms": re.load_global_coord_xforms_um(tile=1),
        "global_xforms_missing": re.load_global_coord_xforms_um(tile=0),
        "chromatic_0.665": re.load_chromatic_affine_transform_zyx_um(wavelength_um=0.665),
        "chromatic_0.59": re.load_chromatic_affine_transform_zyx_um(wavelength_um=0.59),
        "chromatic_missing": re.load_chromatic_affine_transform_zyx_um(wavelength_um=0.7),
        "norm_global": re.load_decode_normalization_vectors(None, "global"),
        "norm_iterative_runA": re.load_decode_normalization_vectors("runA", "iterative"),
        "norm_missing": re.load_decode_normalization_vectors(None, "iterative"),
        "norm_metadata_global": re.load_decode_normalization_metadata(None, "global"),
        "norm_metadata_runA": re.load_decode_normalization_metadata("runA", "iterative"),
        "decoded_tile1": re.load_local_decoded_spots(tile=1),
        "decoded_tile0_runA": re.load_local_decoded_spots(tile=0, decode_run_key="runA"),
        "filtered": re.load_global_filtered_decoded_spots(),
        "decoded_temporary_dir": str(Path(re.decoded_temporary_dir()).relative_to(root)),
        "decoded_temporary_dir_runA": str(Path(re.decoded_temporary_dir(decode_run_key="runA")).relative_to(root)),
    }
    out = {"files": files, "tables": tables, "loads": jsonable(loads)}
    (here / "qi2lab_sidecars.json").write_text(json.dumps(out, indent=1, sort_keys=True))
    print(sorted(files), sorted(tables))
    shutil.rmtree(root.parent, ignore_errors=True)


if __name__ == "__main__":
    main()
