"""CPU: the simulation converters against a run of the REFERENCE's own ``convert_simulation`` and ``convert_data``
(cli/statphysbio_simulation/convert_simulation_to_experiment.py:26-150, convert_to_datastore.py:185-573;
tests/golden/make_golden_simulation.py -> tests/golden/simulation_conversion.npz): the acquisition folder file by file,
then every image, PSF and JSON sidecar of the datastore tree - with the defaults and with ``z_step=2`` + synthetic
chromatic aberration (the resampled readouts go through the package's warp, so that case needs the GPU)."""

import json
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN, pkg


@pytest.fixture(scope="module")
def fixture():
    return np.load(GOLDEN / "simulation_conversion.npz", allow_pickle=False)


def _materialise_simulation(g, root: Path):
    sim = pkg("simulation")
    root.mkdir(parents=True, exist_ok=True)
    sim.write_tiff(root / "aligned_1.tiff", g["in_pages"])
    for name in ("scan_metadata.csv", "codebook.csv", "bit_order.csv"):
        (root / name).write_text(str(g["in_text_" + name]))
    return root


def _materialise_acquisition(g, acq: Path):
    sim = pkg("simulation")
    for rel in [str(x) for x in g["acq_files"]]:
        p = acq / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        if rel.endswith(".tif"):
            sim.write_tiff(p, g["acq_tif_" + rel])
        else:
            p.write_text(str(g["acq_text_" + rel]))
    return acq


def test_convert_simulation_writes_the_reference_runs_acquisition(fixture, tmp_path):
    sim = pkg("simulation")
    g = fixture
    root = _materialise_simulation(g, tmp_path / "simulation")
    acq = sim.convert_simulation(root)
    mine = sorted(str(p.relative_to(acq)) for p in acq.rglob("*") if p.is_file())
    assert mine == [str(x) for x in g["acq_files"]]
    for rel in mine:
        if rel.endswith(".tif"):
            got, want = sim.read_tiff(acq / rel), g["acq_tif_" + rel]
            assert got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want), rel
        else:
            assert (acq / rel).read_text() == str(g["acq_text_" + rel]), rel


def _same_json(a, b, path=""):
    if isinstance(b, dict):
        assert isinstance(a, dict) and sorted(a) == sorted(b), (path, sorted(a) if isinstance(a, dict) else a, sorted(b))
        for k in b:
            _same_json(a[k], b[k], f"{path}/{k}")
    elif isinstance(b, list):
        assert isinstance(a, list) and len(a) == len(b), (path, a, b)
        for i, (x, y) in enumerate(zip(a, b)):
            _same_json(x, y, f"{path}[{i}]")
    elif isinstance(b, float) and not isinstance(b, bool):
        assert isinstance(a, (int, float)) and abs(float(a) - b) <= 1e-7 * max(1.0, abs(b)), (path, a, b)
    else:
        assert a == b and type(a) is type(b), (path, a, b)


def _check_store(g, tag, store_path: Path, store):
    want_json = json.loads(str(g[f"{tag}_json"]))
    mine = {str(p.relative_to(store_path)): json.loads(p.read_text()) for p in store_path.rglob("*.json")
            if ".ome.zarr" not in str(p) and "zarr.json" != p.name}
    assert sorted(mine) == sorted(want_json), (sorted(set(mine) ^ set(want_json)))
    for rel in sorted(want_json):
        _same_json(mine[rel], want_json[rel], rel)
    # images: every corrected image the reference stored, by entity
    for rel in [str(x) for x in g[f"{tag}_images"]]:
        want = g[f"{tag}_image_{rel}"]
        parts = Path(rel).parts
        if parts[0] == "fiducial":
            got = store.load_local_corrected_image(tile=parts[1], round=parts[2], return_future=False)
        elif parts[0] == "readouts":
            got = store.load_local_corrected_image(tile=parts[1], bit=parts[2], return_future=False)
        else:
            continue   # calibration images: PSFs are compared below, the all-zero noise map is outside the hot path
        got = got.cpu().numpy() if hasattr(got, "cpu") else np.asarray(got)
        assert got.dtype == want.dtype and got.shape == want.shape, (rel, got.dtype, got.shape, want.shape)
        assert np.array_equal(got, want), rel
    psfs = np.asarray([np.asarray(p) for p in store.channel_psfs], dtype=np.float32)
    np.testing.assert_array_equal(psfs, g[f"{tag}_psfs"])


def test_convert_data_writes_the_reference_runs_datastore(fixture, tmp_path):
    sim, qs = pkg("simulation"), pkg("qi2lab_store")
    g = fixture
    acq = _materialise_acquisition(g, tmp_path / "sim_acquisition")
    path = tmp_path / "store_default"
    store = sim.convert_data(acq, datastore_factory=lambda **kw: qs.Qi2labDiskStore(path, device=None, **kw))
    _check_store(g, "default", path, store)
    # and what a reader that re-opens the directory sees
    again = qs.Qi2labDiskStore(path)
    assert again.load_local_round_linker(tile=0, bit=5) == 3 and list(again.load_local_bit_linker(tile=0, round=2)) == [5, 6]
    assert again.datastore_state["Corrected"] and again.datastore_state["Calibrations"]


@pytest.mark.gpu
def test_convert_data_with_z_step_and_synthetic_chromatic_aberration(fixture, tmp_path):
    """z_step = 2 halves the planes and doubles dz (PSFs included); the synthetic chromatic aberration resamples every
    readout image through the package's affine warp (HIP) where the reference used its own
    ``warp_array_to_reference_gpu`` (SciPy stand-in in the fixture run): identical uint16 volumes."""
    sim, qs = pkg("simulation"), pkg("qi2lab_store")
    g = fixture
    acq = _materialise_acquisition(g, tmp_path / "sim_acquisition")
    path = tmp_path / "store_zstep2"
    store = sim.convert_data(acq, datastore_factory=lambda **kw: qs.Qi2labDiskStore(path, device="cuda:0", **kw),
                             z_step=2, synthetic_chromatic_aberration=True, synthetic_chromatic_aberration_scale=2.0)
    _check_store(g, "zstep2_chromatic", path, store)
