"""GPU: the simulation matrix harness (the reference's only end-to-end acceptance,
tests/test_simulation_example_pipeline.py:1244-1318): a dataset in the statphysbio on-disk format goes through
convert_simulation -> convert_data -> DataRegistration (decon + phase-correlation registration, with and without the
SOFIMA residual flow) -> optimize_normalization_by_decoding -> decode_all_tiles -> calculate_F1, and the deformable
run must not score below the affine-only run (F1 rounded to 3 decimals, as the reference compares them)."""

import copy

import numpy as np
import pandas as pd
import pytest
from scipy.ndimage import gaussian_filter

from conftest import pkg

pytestmark = pytest.mark.gpu
# Lateral shifts of 5-9 px on purpose.  phase_cross_correlation(disambiguate=True) - the reference's call
# (multiview_registration.py:266-271), restated faithfully - decides between a shift and its wrapped twins by the
# correlation coefficient over the overlap tiles, and accepts tiles of as few as three pixels: with shifts that round
# to 1-4 px the corner tile holds 3-12 samples whose coefficient exceeds the true tile's 0.8-0.9 about one time in
# ten (that did flip one round of an earlier, smaller shift set by 128 px).  At 30-80 samples it cannot.
SHIFTS = [(0, 0, 0), (0.6, 6.4, -7.1), (-0.4, -7.7, 6.2), (0.3, 8.6, 5.8), (0.0, -6.9, -8.4), (0.7, 5.2, 9.3),
          (-0.5, -8.8, -5.6), (0.2, 6.9, -7.7)]


def _predictor(image):
    """Stand-in for U-FISH (third party, out of scope): difference-of-Gaussians spot likelihood in [0, 1] and its
    local maxima as the localisation table."""
    img = np.asarray(image, dtype=np.float32)
    dog = gaussian_filter(img, (1.0, 1.2, 1.2)) - gaussian_filter(img, (2.5, 3.5, 3.5))
    dog = np.clip(dog, 0, None)
    prob = np.clip(dog / max(float(np.percentile(dog, 99.9)), 1e-6), 0.0, 1.0).astype(np.float32)
    flat = np.argsort(prob.ravel())[-20:]
    z, y, x = np.unravel_index(flat, prob.shape)
    return pd.DataFrame({"axis-0": z, "axis-1": y, "axis-2": x}), prob


def test_simulation_f1_sofima_not_below_affine(tmp_path):
    sim = pkg("simulation")
    DR = pkg("DataRegistration")
    PD = pkg("PixelDecoder")
    root = sim.make_simulation(tmp_path / "sim", shape_zyx=(16, 128, 128), n_rna=260, n_beads=220,
                               round_shifts_px=SHIFTS, deformation_px=1.2, seed=5)
    acq = sim.convert_simulation(root)
    base = sim.convert_data(acq)
    gt = pd.read_csv(root / "GT_spots.csv")
    t = base.tile_ids[0]
    base.set_global_coord_xforms_um(t, np.eye(4, dtype=np.float32), np.zeros(3, dtype=np.float32),
                                    np.asarray(base.voxel_size_zyx_um, dtype=np.float32))
    kw = dict(minimum_pixels=9, magnitude_threshold=(0.9, 10.0))
    scores, shifts_found = {}, {}
    for mode, deformable in (("affine", False), ("sofima", True)):
        store = copy.deepcopy(base)
        reg = DR.DataRegistration(store, decon_fiducial=True, decon_readout=True,
                                  perform_deformable_registration=deformable, num_gpus=1, verbose=0,
                                  feature_predictor=_predictor)
        reg.register_all_tiles()
        spacing = np.asarray(store.voxel_size_zyx_um)
        shifts_found[mode] = np.asarray([store.load_local_round_transform_zyx_um(t, r)[:3, 3] / spacing
                                         for r in store.round_ids])
        dec = PD.PixelDecoder(store, merfish_bits=16, num_gpus=1, verbose=0)
        dec.optimize_normalization_by_decoding(n_random_tiles=1, n_iterations=3, **kw)
        dec.decode_all_tiles(assign_to_cells=False, normalization_method="iterative", **kw)
        res = sim.calculate_F1(store, gt, search_radius=1.0)
        scores[mode] = res
        print(f"\\n{mode}: {res}")
        print(np.array2string(shifts_found[mode], precision=2, suppress_small=True))
    # registration recovered the planted round shifts (pull = + shift) to within the smooth warp that rides on them
    # (amplitude 1.2 px: the rigid estimate locks onto the dominant displacement, the residual is SOFIMA's job)
    np.testing.assert_allclose(shifts_found["affine"], np.asarray(SHIFTS), atol=1.2 + 0.35)
    f1a, f1s = scores["affine"]["F1 Score"], scores["sofima"]["F1 Score"]
    assert f1a > 0.6, scores
    assert round(f1s, 3) >= round(f1a, 3), scores   # tests/test_simulation_example_pipeline.py:1308-1318
