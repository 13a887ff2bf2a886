"""CPU: the on-disk qi2lab datastore surface (qi2lab_store.Qi2labDiskStore) against a tree that the reference's
own qi2labDataStore class wrote and read back (tests/golden/qi2lab_sidecars.json, made by
tests/golden/make_golden_datastore.py) - both directions: we read what the reference wrote, and what we write is,
file by file, what the reference wrote for the same calls."""

import json
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

from conftest import ROOT, pkg

FIXTURE = json.loads((ROOT / "tests" / "golden" / "qi2lab_sidecars.json").read_text())


def _frame(spec):
    return pd.DataFrame(spec["records"], columns=spec["columns"])


def _materialise(root: Path):
    for rel, content in FIXTURE["files"].items():
        p = root / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(json.dumps(content))
    for rel, spec in FIXTURE["tables"].items():
        p = root / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        _frame(spec).to_parquet(p, index=False)


def _same(a, b, tol=1e-6):
    if isinstance(b, dict):
        return isinstance(a, dict) and set(a) == set(b) and all(_same(a[k], b[k], tol) for k in b)
    if isinstance(b, (list, tuple)):
        a = a.tolist() if isinstance(a, np.ndarray) else a
        return isinstance(a, (list, tuple)) and len(a) == len(b) and all(_same(x, y, tol) for x, y in zip(a, b))
    if isinstance(b, float) or isinstance(a, (float, np.floating)):
        return a is not None and abs(float(a) - float(b)) <= tol * max(1.0, abs(float(b)))
    if isinstance(a, np.generic):
        a = a.item()
    return a == b


def test_reads_a_tree_written_by_the_reference(tmp_path):
    qs = pkg("qi2lab_store")
    _materialise(tmp_path / "store")
    ds = qs.Qi2labDiskStore(tmp_path / "store")
    L = FIXTURE["loads"]
    assert list(ds.tile_ids) == L["tile_ids"] and list(ds.round_ids) == L["round_ids"] and list(ds.bit_ids) == L["bit_ids"]
    assert (ds.num_bits, ds.num_rounds, ds.num_tiles, ds.microscope_type) == (L["num_bits"], L["num_rounds"], L["num_tiles"], L["microscope_type"])
    assert _same(list(ds.voxel_size_zyx_um), L["voxel_size_zyx_um"])
    assert ds.codebook.equals(_frame(L["codebook"]))
    assert ds.datastore_state == L["datastore_state"]
    for b in range(4):
        assert ds.load_local_round_linker(tile=1, bit=b) == L["round_linker"][f"1/{b}"]
        assert _same(list(ds.load_local_wavelengths_um(tile=1, bit=b)), L["wavelengths_bit"][f"1/{b}"])
    for r in range(2):
        assert ds.load_local_bit_linker(tile=1, round=r) == L["bit_linker"][f"1/{r}"]
        assert _same(list(ds.load_local_wavelengths_um(tile=1, round=r)), L["wavelengths_round"][f"1/{r}"])
        T = ds.load_local_round_transform_zyx_um(tile=1, round=r)
        assert T.dtype == np.float32 and _same(T, L["round_transform"][f"1/{r}"])
        stage, c2s = ds.load_local_stage_position_zyx_um(tile=1, round=r)
        assert stage.dtype == np.float32 and _same([stage, c2s], L["stage"][f"1/{r}"])
    assert ds.load_local_round_transform_zyx_um(tile=0, round=1) is None and L["round_transform_missing"] is None
    assert _same(list(ds.load_global_coord_xforms_um(tile=1)), L["global_xforms"])
    assert list(ds.load_global_coord_xforms_um(tile=0)) == [None, None, None] == L["global_xforms_missing"]
    assert _same(ds.load_chromatic_affine_transform_zyx_um(wavelength_um=0.665), L["chromatic_0.665"])
    assert _same(ds.load_chromatic_affine_transform_zyx_um(wavelength_um=0.59), L["chromatic_0.59"])
    assert _same(ds.load_chromatic_affine_transform_zyx_um(wavelength_um=0.7), L["chromatic_missing"])
    assert _same(list(ds.load_decode_normalization_vectors(None, "global")), L["norm_global"])
    assert _same(list(ds.load_decode_normalization_vectors("runA", "iterative")), L["norm_iterative_runA"])
    assert list(ds.load_decode_normalization_vectors(None, "iterative")) == [None, None] == L["norm_missing"]
    assert ds.load_decode_normalization_metadata(None, "global") is None and L["norm_metadata_global"] is None
    assert ds.load_decode_normalization_metadata("runA", "iterative") == L["norm_metadata_runA"]
    assert ds.load_local_decoded_spots(tile=1).equals(_frame(L["decoded_tile1"]))
    assert ds.load_local_decoded_spots(tile=0, decode_run_key="runA").equals(_frame(L["decoded_tile0_runA"]))
    assert ds.load_local_decoded_spots(tile=0) is None
    assert ds.load_global_filtered_decoded_spots().equals(_frame(L["filtered"]))
    assert str(ds.decoded_temporary_dir().relative_to(tmp_path / "store")) == L["decoded_temporary_dir"]
    assert str(ds.decoded_temporary_dir(decode_run_key="runA").relative_to(tmp_path / "store")) == L["decoded_temporary_dir_runA"]
    assert ds.has_feature_predictor_spots(1, 2) and not ds.has_feature_predictor_spots(1, 1)
    with pytest.raises(ValueError):
        ds.decoded_temporary_dir(decode_run_key="../escape")


def test_writes_the_files_the_reference_writes(tmp_path):
    """The calls of make_golden_datastore.py, made on this package's store."""
    qs = pkg("qi2lab_store")
    cb = _frame(FIXTURE["loads"]["codebook"])
    ds = qs.Qi2labDiskStore(tmp_path / "store", num_tiles=2, num_rounds=2, num_bits=4, codebook=cb,
                            voxel_size_zyx_um=[0.315, 0.098, 0.098], microscope_type="3D",
                            experiment_order=np.asarray([[1, 1, 2], [2, 3, 4]]),
                            channels_in_data=["alexa488", "atto565", "alexa647"], camera_model="flir",
                            tile_overlap=0.2, binning=1, na=1.35, ri=1.51, e_per_ADU=0.54)
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
    rng = np.random.default_rng(0)
    nrm = rng.uniform(100, 300, size=4).astype(np.float32)
    bkd = rng.uniform(1, 20, size=4).astype(np.float32)
    ds.save_decode_normalization_vectors(None, "global", nrm, bkd, decode_mode="3d")
    ds.save_decode_normalization_vectors("runA", "iterative", nrm * 2, bkd * 2, decode_mode="3d",
                                         metadata={"codebook_fingerprint": "abc", "n_iterations": 3})
    spots = _frame(FIXTURE["loads"]["decoded_tile1"])
    ds.save_local_decoded_spots(spots, tile=1)
    ds.save_local_decoded_spots(spots.iloc[:1], tile=0, decode_run_key="runA")
    ds.save_global_filtered_decoded_spots(spots)
    ds.save_local_feature_predictor_spots(pd.DataFrame({"z": [1], "y": [2], "x": [3]}), tile=1, bit=2)
    state = ds.datastore_state
    state.update({"Calibrations": True, "Corrected": True})
    ds.datastore_state = state

    root = tmp_path / "store"
    ours = {str(p.relative_to(root)): json.loads(p.read_text()) for p in root.rglob("*.json")}
    assert set(ours) == set(FIXTURE["files"])
    for rel, want in FIXTURE["files"].items():
        assert _same(ours[rel], want), rel
    tables = {str(p.relative_to(root)) for p in root.rglob("*.parquet")}
    assert tables == set(FIXTURE["tables"])
    for rel, spec in FIXTURE["tables"].items():
        got = pd.read_parquet(root / rel)
        assert list(got.columns) == spec["columns"] and _same(got.to_numpy().tolist(), spec["records"]), rel
    assert (root / "all_tiles_filtered_decoded_features" / "decoded_features.csv.gz").exists()
    # and the tree re-opens
    again = qs.Qi2labDiskStore(root)
    assert again.load_local_round_linker(1, 3) == 2 and again.codebook.equals(cb)
    with pytest.raises(ValueError):
        qs.Qi2labDiskStore(root, num_tiles=1)


def test_operators_see_the_in_memory_helpers_too(tmp_path):
    """set_* helpers used by tests / bench map onto the reference's save_* methods."""
    qs = pkg("qi2lab_store")
    cb = _frame(FIXTURE["loads"]["codebook"])
    ds = qs.Qi2labDiskStore(tmp_path / "s", num_tiles=1, num_rounds=2, num_bits=4, codebook=cb)
    img = np.random.default_rng(0).integers(90, 300, size=(4, 20, 24)).astype(np.uint16)
    ds.set_corrected_image(img, tile=0, bit=1, wavelengths_um=(0.635, 0.67))
    assert ds.load_local_wavelengths_um(0, bit=1) == (0.635, 0.67)
    assert np.array_equal(ds.load_local_readout_image(0, bit=1, return_future=False), img)
    ds.set_round_linker(0, 1, 2)
    assert ds.load_local_round_linker(0, 1) == 2
    ds.set_stage_position_zyx_um(0, (1.0, 2.0, 3.0))
    stage, c2s = ds.load_local_stage_position_zyx_um(0, round=0)
    assert stage.tolist() == [1.0, 2.0, 3.0] and np.array_equal(c2s, np.eye(4, dtype=np.float32))
    ds.set_global_coord_xforms_um(0, np.eye(4), (0, 0, 0), (0.3, 0.1, 0.1))
    assert ds.load_global_coord_xforms_um(0)[2].tolist() == pytest.approx([0.3, 0.1, 0.1])
    ds.set_chromatic_affine_transform_zyx_um(0.67, np.eye(4) * 2)
    assert ds.load_chromatic_affine_transform_zyx_um(wavelength_um=0.67)[0, 0] == 2.0
    ds.save_local_sofima_flow_field(np.zeros((3, 1, 2, 2), np.float32), tile=0, round=1, sofima_status="ok")
    assert ds.load_local_sofima_flow_field(0, 1)[1] == {"sofima_status": "ok"}
    # image-level attributes are part of the entity's metadata (reference: _load_entity_attributes)
    assert ds._load_entity_attributes(ds._fiducial_root_path / "tile0000" / "round002")["sofima_status"] == "ok"


def test_channel_psfs_live_in_the_calibrations_folder(tmp_path):
    qs = pkg("qi2lab_store")
    cb = _frame(FIXTURE["loads"]["codebook"])
    ds = qs.Qi2labDiskStore(tmp_path / "s", num_tiles=1, num_rounds=1, num_bits=4, codebook=cb)
    assert ds.channel_psfs is None
    rng = np.random.default_rng(1)
    psfs = rng.uniform(size=(3, 5, 7, 7)).astype(np.float32)
    ds.channel_psfs = psfs
    manifest = json.loads((tmp_path / "s" / "calibrations" / "attributes.json").read_text())["psf_manifest"]
    assert manifest == {str(i): {"id": f"psf_{i:03d}", "shape_zyx": [5, 7, 7]} for i in range(3)}   # :736-741
    assert (tmp_path / "s" / "calibrations" / "psf_data" / "psf_002.ome.zarr" / "0" / "zarr.json").exists()
    again = qs.Qi2labDiskStore(tmp_path / "s")
    assert np.array_equal(again.channel_psfs, psfs)
    ragged = [psfs[0], psfs[1][:, :5, :5]]
    again.channel_psfs = ragged
    got = qs.Qi2labDiskStore(tmp_path / "s").channel_psfs
    assert isinstance(got, list) and len(got) == 2 and np.array_equal(got[1], ragged[1])
    with pytest.raises(ValueError):
        again.channel_psfs = []


def test_store_survives_pickling_into_a_worker(tmp_path):
    import pickle

    qs = pkg("qi2lab_store")
    cb = _frame(FIXTURE["loads"]["codebook"])
    ds = qs.Qi2labDiskStore(tmp_path / "s", num_tiles=1, num_rounds=1, num_bits=4, codebook=cb, num_threads=2)
    img = np.random.default_rng(0).integers(90, 300, size=(4, 20, 24)).astype(np.uint16)
    ds.set_corrected_image(img, tile=0, round=0)
    clone = pickle.loads(pickle.dumps(ds))
    assert clone._io is not ds._io and clone._io.num_threads == 2
    assert np.array_equal(clone.load_local_corrected_image(0, round=0, return_future=False), img)
    clone.save_local_round_transform_zyx_um(np.eye(4), tile=0, round=0)
    assert ds.load_local_round_transform_zyx_um(0, 0) is not None            # same files
