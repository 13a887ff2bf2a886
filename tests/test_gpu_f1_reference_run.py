"""GPU: the north_star's F1 clause against a run of the REFERENCE, not of the oracle.

tests/golden/f1_reference_run.json (made by tests/golden/make_golden_f1_reference.py in the authoring container) is
the dz = 0.315 um cell of the simulation matrix taken through the reference's own ``chunked_rlgc`` (every volume),
``PixelDecoder.optimize_normalization_by_decoding``, ``decode_all_tiles`` (``decode_one_tile`` / ``_extract_barcodes``
/ blank-fraction filter) and the scorer CLI's ``calculate_F1`` under the stand-ins of make_golden.py; the round
transforms it decoded with are in the fixture.  Here the same dataset is regenerated from its seed, the same
transforms are written to the store, and the HIP pipeline does the rest:

    |F1_hip - F1_reference| <= 0.005       (BASELINE.json north_star: "within 0.5 % of reference")

with the HIP solver fed the photon splits the reference run drew (NumPy's stream, replayed by the oracle - which is
the reference's run bit for bit - and injected), so that the comparison is of arithmetic, not of two realisations of a
random split; the build's own Philox stream is then held to the oracle pipeline on the same stream.
Reference: tests/test_simulation_example_pipeline.py:1244-1318, PixelDecoder.py:4471-4868,
cli/statphysbio_simulation/calculate_F1.py:208-276."""

import json

import numpy as np
import pytest

import oracle_pipeline as op
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _cells():
    return sorted(json.loads((GOLDEN / "f1_reference_run.json").read_text()))


@pytest.mark.parametrize("name", _cells())
def test_hip_pipeline_f1_stands_with_the_reference_run(name, tmp_path):
    fx = json.loads((GOLDEN / "f1_reference_run.json").read_text())[name]
    assert fx["cell"] == name
    cell = op.CELLS[fx["cell"]]
    assert fx["seed"] == cell["seed"] and fx["shifts"] == [list(s) for s in cell["shifts"]]
    transforms = [np.asarray(T) for T in fx["round_transforms_zyx_um"]]
    want = fx["f1"]

    # (a) the reference's own photon splits: every stage deterministic and comparable
    store, gt = op.make_cell(tmp_path / "a", fx["cell"])
    same = op.hip_readouts_then_score(store, gt, transforms, float(fx["radius"]), dict(cell["decode"]),
                                      reference_stream=True)
    print(f"\nHIP, reference stream {same}\nreference run         {want} (before / after filter "
          f"{fx['decoded_before_filter']} / {fx['decoded_after_filter']})")
    assert same["uint16_flips"] < 5e-3
    assert abs(same["F1 Score"] - want["F1 Score"]) <= 0.005, (same, want)
    # before the blank-fraction filter the two tables agree to a row or two (4913 / 4912 in round 4); the filter's
    # threshold sweep is discrete and moves ~1 % of the rows for a last-digit change of its inputs - the reference's own
    # sensitivity, visible in precision more than in F1
    assert abs(same["decoded_before_filter"] - fx["decoded_before_filter"]) <= 0.002 * fx["decoded_before_filter"] + 2
    assert abs(same["decoded_after_filter"] - fx["decoded_after_filter"]) <= 0.02 * fx["decoded_after_filter"]
    assert abs(same["Precision"] - want["Precision"]) <= 0.01 and abs(same["Recall"] - want["Recall"]) <= 0.01
    np.testing.assert_allclose(same["normalization"], fx["normalization"], rtol=0.05)

    # (b) the build's own Philox stream: a different realisation of the random split.  It stands with the ORACLE
    # pipeline on the same stream and transforms (fixture: oracle_pipeline_same_transforms) to the north_star's bound;
    # against the reference run's NumPy stream the two realisations differ by 0.0065 in F1 on this dataset
    # (0.7979 vs 0.7914: the blank-fraction sweep lands on another threshold) - the draw, not the arithmetic.
    store, gt = op.make_cell(tmp_path / "b", fx["cell"])
    own = op.hip_readouts_then_score(store, gt, transforms, float(fx["radius"]), dict(cell["decode"]))
    twin = fx["oracle_pipeline_same_transforms"]
    print(f"HIP, Philox stream    {own}\noracle, Philox stream {twin}")
    assert abs(own["F1 Score"] - twin["F1 Score"]) <= 0.005, (own, twin)
    assert abs(own["F1 Score"] - want["F1 Score"]) <= 0.015, (own, want)
