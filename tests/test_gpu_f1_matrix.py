"""GPU: the simulation F1 matrix (tests/test_simulation_example_pipeline.py:48-52, 1244-1318 of the reference): three
axial spacings (0.315 / 1.0 / 1.5 um; F1 radii 1.0 / 1.0 / 1.5 um) x {affine only, affine + SOFIMA residual flow} on
16 x 256 x 256 datasets in the statphysbio layout.  Each cell goes through the package's operators on the GPU
(convert -> DataRegistration -> optimize_normalization_by_decoding -> decode_all_tiles -> calculate_F1) and stands next to
the ORACLE pipeline's F1 for the same dataset (tests/golden/f1_matrix.json, computed once on the CPU by
tests/golden/make_golden_f1_matrix.py from the same seeds):

    |F1_hip - F1_oracle| <= 0.005      north_star: "decoded-spot F1 within 0.5 % of reference on the simulation matrix"
    F1(SOFIMA) >= F1(affine) to 3 dp   the reference's own acceptance criterion (:1308-1318)

The first cell's round shifts all round to 1-4 px - the range bench config 3 draws, where
``phase_cross_correlation(disambiguate=True)`` meets its three-sample corner tiles: the registration must land within
the reference's +-0.35 px (+ the planted 1.2 px warp) there too."""

import copy
import json

import numpy as np
import pytest

import oracle_pipeline as op
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cell", list(op.CELLS))
def test_f1_of_the_hip_pipeline_stands_with_the_oracle_pipelines(cell, tmp_path):
    fixture = json.loads((GOLDEN / "f1_matrix.json").read_text())
    want = fixture[cell]
    meta = fixture["_meta"]["cells"][cell]
    assert meta["seed"] == op.CELLS[cell]["seed"] and meta["shifts"] == [list(s) for s in op.CELLS[cell]["shifts"]]
    base, gt = op.make_cell(tmp_path, cell)
    scores = {}
    for mode, deformable in (("affine", False), ("sofima", True)):
        store = copy.deepcopy(base)
        scores[mode] = op.hip_pipeline_score(store, gt, op.CELLS[cell]["radius"], deformable, op.CELLS[cell]["decode"])
        if mode == "affine":
            t, sp = store.tile_ids[0], np.asarray(store.voxel_size_zyx_um)
            found = np.asarray([np.asarray(store.load_local_round_transform_zyx_um(t, r))[:3, 3] / sp
                                for r in store.round_ids])
            # the planted shifts (pull = + shift), to the reference's registration tolerance plus the smooth warp that
            # rides on them; and the oracle's own estimates to the HIP-vs-oracle bound of tests/test_gpu_registration.py
            np.testing.assert_allclose(found, np.asarray(op.CELLS[cell]["shifts"]), atol=1.2 + 0.35)
            np.testing.assert_allclose(found, np.asarray(want["shifts_found_px"]), atol=0.05)
        print(f"\n{cell} {mode}: HIP {scores[mode]}\n{' ' * (len(cell) + len(mode) + 2)}oracle {want[mode]}")
    for mode in ("affine", "sofima"):
        assert abs(scores[mode]["F1 Score"] - want[mode]["F1 Score"]) <= 0.005, (cell, mode, scores[mode], want[mode])
    assert round(scores["sofima"]["F1 Score"], 3) >= round(scores["affine"]["F1 Score"], 3), scores
    assert round(want["sofima"]["F1 Score"], 3) >= round(want["affine"]["F1 Score"], 3), want
