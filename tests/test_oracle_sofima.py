"""CPU: the flow-estimator oracle against the reference-run glue fixtures and the reference's
own tolerance test (tests/test_sofima_deformable_registration.py:221-527)."""

import numpy as np

import sofima_synth as SY
from oracle import sofima_ref as S

CFG = S.SofimaRegistrationConfig()


def test_glue_matches_reference_run(golden):
    g = golden("sofima_glue")
    shapes = [(16, 256, 256), (64, 2048, 2048), (3, 20, 500), (31, 64, 64)]
    assert np.asarray([S.resolve_patch_and_step(s, CFG) for s in shapes]).tolist() == g["patch_step"].tolist()
    st, md = S.stabilize_axial_flow_component(g["flow"], CFG)
    np.testing.assert_array_equal(st, g["stabilized"])
    assert md["axial_flow_median_px"] == float(g["stab_median"])
    assert md["axial_flow_clipped_vectors"] == int(g["stab_clipped"])
    assert md["axial_flow_valid_vectors"] == int(g["stab_valid"])
    np.testing.assert_array_equal(S.median_initial_flow_field(g["sparse_flow"]), g["median_init"])
    assert S.normalized_mean_squared_error(g["nmse_a"], g["nmse_b"]) == float(g["nmse"])


def test_compose_and_subpixel_match_reference_run(golden):
    g = golden("sofima_glue_jax")
    c = S.compose_flow_fields_same_grid(g["base"], g["residual"], tuple(g["stride"]), tuple(g["box_start"]))
    np.testing.assert_array_equal(c, g["composed"])
    r, n = S.refine_flow_vectors_subpixel(g["flow"], g["moving"], g["fixed"], tuple(int(v) for v in g["patch"]),
                                          tuple(int(v) for v in g["step"]), CFG)
    assert n == int(g["refined_count"])
    np.testing.assert_array_equal(r, g["refined"])


def test_patch_xcorr_is_the_direct_sum():
    rng = np.random.default_rng(3)
    a = rng.normal(size=(3, 5, 4)).astype(np.float32)
    b = rng.normal(size=(3, 5, 4)).astype(np.float32)
    full = S.patch_xcorr_full(a, b)
    ac, bc = a - a.mean(), b - b.mean()
    direct = np.zeros_like(full)
    for dz in range(-2, 3):
        for dy in range(-4, 5):
            for dx in range(-3, 4):
                s = 0.0
                for u in np.ndindex(3, 5, 4):
                    w = (u[0] + dz, u[1] + dy, u[2] + dx)
                    if all(0 <= w[i] < a.shape[i] for i in range(3)):
                        s += float(bc[u]) * float(ac[w])
                direct[dz + 2, dy + 4, dx + 3] = s
    np.testing.assert_allclose(full, direct, atol=2e-5)


def test_peak_stats_and_clean_flow_semantics():
    full = np.zeros((5, 21, 21), dtype=np.float32)
    full[2, 10, 12] = 10.0   # main peak -> d = (+2 x, 0 y, 0 z) for patch (3, 11, 11)
    full[2, 10, 13] = -4.0   # minimum inside the radius
    full[2, 2, 2] = 6.0      # second local maximum outside radius 3
    full[2, 2, 3] = 5.5      # suppressed neighbour
    pk, sharp, ratio = S.peak_stats(full, min_distance=2, peak_radius=3)
    assert pk == (2, 10, 12) and sharp == np.float32(-2.5) and ratio == np.float32(10.0 / 6.0)
    full[2, 2, 2] = 4.0      # below the 0.5 relative threshold -> no second peak
    full[2, 2, 3] = 0.0
    assert S.peak_stats(full, 2, 3)[2] == 0.0
    flow = np.zeros((5, 3, 3, 3), dtype=np.float32)
    flow[3] = -2.5
    flow[4] = 0.0
    flow[3, 0, 0, 0] = 1.1          # weak sharpness
    flow[4, 0, 0, 1] = 1.1          # ambiguous second peak
    flow[0, 0, 0, 2] = 31.0         # too large
    flow[1, 2, 2, 2] = 6.0          # local outlier
    c = S.clean_flow(flow, 1.2, 1.2, 30.0, 5.0)
    bad = np.isnan(c[0])
    assert bad.sum() == 4 and bad[0, 0, 0] and bad[0, 0, 1] and bad[0, 0, 2] and bad[2, 2, 2]


def test_mesh_relaxes_to_the_measurements_and_fills_gaps():
    rng = np.random.default_rng(5)
    grid = (3, 5, 6)
    g = np.indices(grid, dtype=np.float32)
    smooth = np.stack([0.5 * g[2] - 1.0, 0.3 * g[1], 0.2 * g[0]]).astype(np.float32)
    meas = smooth.copy()
    hole = rng.uniform(size=grid) < 0.3
    meas[:, hole] = np.nan
    import dataclasses
    cfg = dataclasses.replace(CFG, mesh_num_iters=500)
    x, e_kin, iters = S.relax_mesh(S.median_initial_flow_field(meas), meas, (16.0, 16.0, 5.0), cfg)
    assert iters < cfg.mesh_max_iters and e_kin[-1] < 1e-4            # stopped on the velocity criterion
    np.testing.assert_allclose(x[:, ~hole], smooth[:, ~hole], atol=0.2)    # k0 >> k pins measured nodes
    np.testing.assert_allclose(x[:, hole], smooth[:, hole], atol=0.8)      # gaps follow the neighbours
    f = S.elastic_mesh_3d(x, np.float32(cfg.mesh_k), (16.0, 16.0, 5.0))
    assert np.abs(f[:, hole]).max() < 1e-3                                 # unmeasured nodes are force free


def test_oracle_estimator_passes_the_reference_tolerance_test():
    fixed, moving, flow = SY.make_pair()
    est, md = S.estimate_sofima_flow_field_xyz_px(fixed, moving)
    SY.check_recovery(est, md, flow)
    assert md["sofima_normalized_mse"] < md["affine_normalized_mse"]
    assert md["map_box_start_xyz_px"] == [16.0, 16.0, 5.0] and md["map_stride_zyx_px"] == [5.0, 16.0, 16.0]
