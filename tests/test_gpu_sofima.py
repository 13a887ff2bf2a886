"""GPU parity: residual flow estimation (utils/sofima_registration.py) against the oracle, the
reference-run glue fixtures and the reference's tolerance test
(tests/test_sofima_deformable_registration.py:221-527)."""

import dataclasses

import numpy as np
import pytest
import torch
from scipy.ndimage import gaussian_filter, map_coordinates

import sofima_synth as SY
from conftest import pkg
from oracle import sofima_ref as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CFG = S.SofimaRegistrationConfig()


def _textured_pair(shape, seed, shift=(1.0, -2.0, 3.0)):
    rng = np.random.default_rng(seed)
    fixed = (gaussian_filter(rng.normal(size=shape), 1.5) * 100 + 50).astype(np.float32)
    g = np.indices(shape, dtype=np.float32)
    wob = 1.5 * np.sin(g[2] / shape[2] * 2 * np.pi)
    moving = map_coordinates(fixed, [g[0] - shift[0], g[1] - shift[1] - wob, g[2] - shift[2]], order=1,
                             mode="nearest").astype(np.float32)
    return fixed, moving


@pytest.mark.parametrize("shape,patch,step,batch", [((14, 40, 44), (6, 12, 12), (3, 6, 6), None),
                                                     ((12, 70, 66), (10, 32, 32), (5, 16, 16), 5),
                                                     ((9, 30, 31), (4, 9, 10), (2, 4, 5), 7)])
def test_patch_flow_field_matches_oracle(shape, patch, step, batch):
    sf = pkg("utils.sofima_registration")
    fixed, moving = _textured_pair(shape, seed=sum(shape))
    ref = S.patch_flow_field(moving, fixed, patch, step, 2, 8)
    got = sf.patch_flow_field(torch.from_numpy(moving).to(DEV), torch.from_numpy(fixed).to(DEV), patch_size=patch,
                              step=step, peak_min_distance=2, peak_radius=8, batch_nodes=batch)
    assert got.shape == ref.shape
    np.testing.assert_array_equal(got[:3], ref[:3])                      # integer peaks
    np.testing.assert_allclose(got[3], ref[3], rtol=2e-3, atol=1e-3)     # sharpness
    np.testing.assert_allclose(got[4], ref[4], rtol=2e-3, atol=1e-3)     # ratio (0 = no second peak)
    assert np.array_equal(got[4] == 0, ref[4] == 0)


def test_fused_cross_correlation_equals_the_rocfft_route():
    """Transform box 24 x 64 x 64 (the configured 10 x 32 x 32 patches): the LDS / register transforms of
    m3d_patch_xcorr_fused against pack -> rocFFT -> cross spectrum -> rocFFT -> peaks on the same volumes."""
    sf = pkg("utils.sofima_registration")
    kernels = pkg("kernels")
    patch, step = (10, 32, 32), (5, 16, 16)
    assert kernels.patch_xcorr_fused_supported(patch, (24, 64, 64))
    assert not kernels.patch_xcorr_fused_supported((6, 12, 12), (12, 24, 24))
    fixed, moving = _textured_pair((26, 150, 178), seed=11, shift=(0.0, 3.0, -2.0))
    pre, post = torch.from_numpy(moving).to(DEV), torch.from_numpy(fixed).to(DEV)
    kw = dict(patch_size=patch, step=step, peak_min_distance=2, peak_radius=8)
    assert sf._XCORR_FUSED
    fused = sf.patch_flow_field(pre, post, batch_nodes=37, **kw)       # ragged last batch
    try:
        sf._XCORR_FUSED = False
        plain = sf.patch_flow_field(pre, post, **kw)
    finally:
        sf._XCORR_FUSED = True
    assert fused.shape == plain.shape == (5, 4, 8, 10)
    np.testing.assert_array_equal(fused[:3], plain[:3])
    np.testing.assert_allclose(fused[3], plain[3], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(fused[4], plain[4], rtol=1e-3, atol=1e-3)
    assert np.abs(fused[:3]).max() > 0


def test_clean_flow_is_the_oracle_rule():
    sf = pkg("utils.sofima_registration")
    rng = np.random.default_rng(2)
    flow = rng.normal(0, 3, size=(5, 4, 6, 7)).astype(np.float32)
    flow[3] = rng.uniform(-4, 4, size=flow.shape[1:])
    flow[4] = np.where(rng.uniform(size=flow.shape[1:]) < 0.5, 0.0, rng.uniform(1.0, 3.0, size=flow.shape[1:]))
    flow[0, 1, 1, 1] = 40.0
    a = sf.clean_flow(flow, min_peak_ratio=1.2, min_peak_sharpness=1.2, max_magnitude=30.0, max_deviation=5.0)
    b = S.clean_flow(flow, 1.2, 1.2, 30.0, 5.0)
    np.testing.assert_array_equal(a, b)


def test_subpixel_refine_and_compose_match_reference_run(golden):
    sf = pkg("utils.sofima_registration")
    g = golden("sofima_glue_jax")
    cfg = sf.SofimaRegistrationConfig()
    got = sf._compose_flow_fields_same_grid(g["base"], g["residual"], stride_zyx=tuple(g["stride"]),
                                            box_start_xyz=tuple(g["box_start"]))
    np.testing.assert_allclose(got, g["composed"], rtol=0, atol=1e-6)
    refined, n = sf._refine_flow_vectors_subpixel(g["flow"], g["moving"], g["fixed"],
                                                  patch_size_zyx=tuple(int(v) for v in g["patch"]),
                                                  step_zyx=tuple(int(v) for v in g["step"]), config=cfg)
    assert n == int(g["refined_count"])
    np.testing.assert_array_equal(np.isnan(refined), np.isnan(g["refined"]))
    same = np.all((refined == g["refined"]) | np.isnan(refined), axis=0)
    assert same.mean() >= 0.97, same.mean()      # float32 SSD vs the double stand-in: near-ties may flip


def test_glue_matches_reference_run(golden):
    sf = pkg("utils.sofima_registration")
    g = golden("sofima_glue")
    cfg = sf.SofimaRegistrationConfig()
    shapes = [(16, 256, 256), (64, 2048, 2048), (3, 20, 500), (31, 64, 64)]
    assert np.asarray([sf._resolve_patch_and_step(s, cfg) for s in shapes]).tolist() == g["patch_step"].tolist()
    st, md = sf._stabilize_axial_flow_component(g["flow"], cfg)
    np.testing.assert_array_equal(st, g["stabilized"])
    assert md["axial_flow_clipped_vectors"] == int(g["stab_clipped"])
    np.testing.assert_array_equal(sf._median_initial_flow_field(g["sparse_flow"]), g["median_init"])
    np.testing.assert_allclose(sf._normalized_mean_squared_error(g["nmse_a"], g["nmse_b"]), float(g["nmse"]),
                               rtol=1e-5)
    assert cfg.as_metadata() == CFG.as_metadata()


@pytest.mark.parametrize("seed", [5, 6])
def test_mesh_relax_matches_oracle(seed):
    sf = pkg("utils.sofima_registration")
    rng = np.random.default_rng(seed)
    grid = (3, 5, 6)
    g = np.indices(grid, dtype=np.float32)
    smooth = np.stack([0.5 * g[2] - 1.0, 0.3 * g[1], 0.2 * g[0]]).astype(np.float32)
    smooth += rng.normal(0, 0.1, size=smooth.shape).astype(np.float32)
    meas = smooth.copy()
    meas[:, rng.uniform(size=grid) < 0.3] = np.nan
    cfg_o = dataclasses.replace(CFG, mesh_num_iters=500)
    cfg = dataclasses.replace(sf.SofimaRegistrationConfig(), mesh_num_iters=500)
    init = S.median_initial_flow_field(meas)
    ref, e_kin, iters = S.relax_mesh(init, meas, (16.0, 16.0, 5.0), cfg_o)
    got, md = sf._relax_flow_field(meas, init, (5, 16, 16), cfg)
    assert md["mesh_iterations"] == iters and md["mesh_relaxation"]
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-3)     # same equilibrium (stop_v_max = 1e-3)
    assert md["mesh_final_kinetic_energy"] < 1e-6


def test_estimator_passes_the_reference_tolerance_test_and_tracks_the_oracle():
    sf = pkg("utils.sofima_registration")
    fixed, moving, flow = SY.make_pair()
    est, md = sf.estimate_sofima_flow_field_xyz_px(fixed, moving)
    assert isinstance(est, np.ndarray) and est.dtype == np.float32
    SY.check_recovery(est, md, flow)
    assert md["sofima_normalized_mse"] < md["affine_normalized_mse"]
    ref, rmd = S.estimate_sofima_flow_field_xyz_px(fixed, moving)
    assert md["valid_flow_vectors"] == rmd["valid_flow_vectors"]
    assert md["subpixel_refined_vectors"] == rmd["subpixel_refined_vectors"]
    np.testing.assert_allclose(md["sofima_normalized_mse"], rmd["sofima_normalized_mse"], rtol=2e-2)
    assert np.sqrt(np.mean((est - ref) ** 2)) < 0.05
    for key in ("map_stride_zyx_px", "map_box_start_xyz_px", "map_box_size_xyz_px", "mesh_initializer", "status"):
        assert md[key] == rmd[key]
    # device tensors in -> same field
    est_t, _ = sf.estimate_sofima_flow_field_xyz_px(torch.from_numpy(fixed).to(DEV), torch.from_numpy(moving).to(DEV))
    np.testing.assert_array_equal(est_t, est)


def test_estimator_fallbacks():
    sf = pkg("utils.sofima_registration")
    flat = np.full((12, 40, 40), 7.0, dtype=np.float32)
    est, md = sf.estimate_sofima_flow_field_xyz_px(flat, flat)
    assert md["status"] in ("identity_fallback_no_valid_vectors", "identity_fallback_no_error_improvement")
    assert not est.any()
    with pytest.raises(ValueError):
        sf.estimate_sofima_flow_field_xyz_px(flat, flat[:, :-1])
