"""GPU: the DataRegistration operator end to end on an in-memory store - deconvolved
fiducials, round transforms, SOFIMA flow fields, per-bit deconvolution with the PSF chosen
by excitation wavelength, feature-predictor tables, resume (DataRegistration.py:390-1121,
1522-1591, 2086-2315)."""

import numpy as np
import pandas as pd
import pytest
from scipy.ndimage import gaussian_filter, shift as ndi_shift

from conftest import pkg

pytestmark = pytest.mark.gpu
SHAPE = (16, 96, 96)
SHIFTS = {"round002": (1.0, 3.0, -2.0), "round003": (0.0, -4.0, 5.0)}


def _psf(sig):
    z, y, x = np.indices((5, 7, 7), dtype=np.float64) - np.asarray([2, 3, 3])[:, None, None, None]
    p = np.exp(-0.5 * ((z / sig[0]) ** 2 + (y / sig[1]) ** 2 + (x / sig[2]) ** 2))
    return (p / p.sum()).astype(np.float32)


def _beads(seed):
    rng = np.random.default_rng(seed)
    vol = np.zeros(SHAPE, dtype=np.float32)
    pts = rng.uniform((3, 10, 10), (SHAPE[0] - 3, SHAPE[1] - 10, SHAPE[2] - 10), size=(60, 3)).astype(int)
    vol[tuple(pts.T)] = rng.uniform(2000, 6000, size=60)
    return gaussian_filter(vol, (1.0, 1.6, 1.6)) * 20 + 100


class _Predictor:
    def __init__(self):
        self.calls = 0

    def __call__(self, image):
        self.calls += 1
        img = np.asarray(image, dtype=np.float32)
        prob = (img / max(float(img.max()), 1.0)).astype(np.float32)
        flat = np.argsort(prob.ravel())[-5:]
        z, y, x = np.unravel_index(flat, prob.shape)
        return pd.DataFrame({"axis-0": z, "axis-1": y, "axis-2": x}), prob


def _store():
    ds_mod = pkg("datastore")
    cb = pd.DataFrame({"gene_id": ["a"], **{f"bit{i + 1:02d}": [1] for i in range(4)}})
    ds = ds_mod.InMemoryDataStore(num_tiles=1, num_rounds=3, num_bits=4, codebook=cb,
                                  voxel_size_zyx_um=(0.31, 0.1, 0.1))
    ds.channel_psfs = [_psf((0.9, 1.2, 1.2)), _psf((1.0, 1.5, 1.5)), _psf((1.2, 1.9, 1.9))]
    fid = _beads(1)
    rng = np.random.default_rng(9)
    t = ds.tile_ids[0]
    ds.set_corrected_image(rng.poisson(fid).astype(np.uint16), tile=t, round="round001")
    for r, s in SHIFTS.items():
        moved = ndi_shift(fid, s, order=1, mode="nearest")
        ds.set_corrected_image(rng.poisson(moved).astype(np.uint16), tile=t, round=r)
    for i, b in enumerate(ds.bit_ids):
        ex = 0.561 if i % 2 == 0 else 0.635
        ds.set_corrected_image(rng.poisson(_beads(10 + i)).astype(np.uint16), tile=t, bit=b,
                               wavelengths_um=(ex, ex + 0.03))
        ds.set_round_linker(t, b, 1 + i // 2)
    return ds


def test_register_all_tiles_end_to_end(capsys):
    dr_mod = pkg("DataRegistration")
    rl = pkg("utils.rlgc")
    ds = _store()
    t = ds.tile_ids[0]
    pred = _Predictor()
    reg = dr_mod.DataRegistration(ds, decon_fiducial=True, decon_readout=True, num_gpus=1, crop_yx_decon=2048,
                                  verbose=0, feature_predictor=pred)
    reg.register_all_tiles()

    # round 1: identity; moving rounds: the known shift (moving = fixed shifted by s => pull transform = -s)
    np.testing.assert_array_equal(ds.load_local_round_transform_zyx_um(t, "round001"), np.eye(4, dtype=np.float32))
    spacing = np.asarray(ds.voxel_size_zyx_um)
    for r, s in SHIFTS.items():
        T = ds.load_local_round_transform_zyx_um(t, r)
        assert T.dtype == np.float32 and T.shape == (4, 4)
        np.testing.assert_allclose(T[:3, :3], np.eye(3), atol=1e-6)
        np.testing.assert_allclose(T[:3, 3] / spacing, s, atol=0.35)

    # deconvolved fiducials: uint16 clip of RLGC with PSF 0, bit-identical to a direct call
    for r in ds.round_ids:
        saved = ds.load_local_fiducial_image(t, r, return_future=False)
        assert saved.dtype == np.uint16 and saved.shape == SHAPE
        raw = ds.load_local_corrected_image(t, round=r, return_future=False)
        direct = rl.chunked_rlgc(raw, ds.channel_psfs[0], gpu_id=0, crop_yx=2048)
        np.testing.assert_array_equal(saved, np.clip(direct, 0, 65535).astype(np.uint16))

    # flow fields for the moving rounds only, with the reference's attributes
    assert ds.load_local_sofima_flow_field(t, "round001") is None
    for r in SHIFTS:
        flow, attrs = ds.load_local_sofima_flow_field(t, r)
        assert flow.dtype == np.float32 and flow.shape[0] == 3
        assert attrs["map_stride_zyx_px"] == [5.0, 16.0, 16.0] and attrs["map_box_start_xyz_px"] == [16.0, 16.0, 5.0]
        assert attrs["reference_round"] == "round001" and attrs["reference_shape_zyx_px"] == SHAPE
        assert attrs["sofima_status"] in ("ok", "identity_fallback_no_error_improvement",
                                          "identity_fallback_no_valid_vectors")
        assert np.abs(flow).max() < 3.0       # the residual after the affine is small

    # readout bits: PSF 1 below 0.600 um excitation, PSF 2 above; tables as the reference builds them
    assert pred.calls == 4
    for i, b in enumerate(ds.bit_ids):
        raw = ds.load_local_corrected_image(t, bit=b, return_future=False)
        direct = rl.chunked_rlgc(raw, ds.channel_psfs[1 if i % 2 == 0 else 2], gpu_id=0, crop_yx=2048)
        want = np.clip(direct, 0, 65535).astype(np.uint16)
        saved = ds.load_local_readout_image(t, b, return_future=False)
        np.testing.assert_array_equal(saved, want)
        prob = ds.load_local_feature_predictor_image(t, b, return_future=False)
        spots = ds._predictor_spots[(t, "bit", b)]
        assert list(spots.columns) == ["z", "y", "x", "sum_prob_pixels", "sum_decon_pixels", "tile_idx", "bit_idx",
                                       "tile_z_px", "tile_y_px", "tile_x_px"]
        assert (spots["bit_idx"] == i + 1).all() and (spots["tile_idx"] == 0).all()
        np.testing.assert_array_equal(spots["tile_y_px"], spots["y"])
        z, y, x = (int(v) for v in spots.iloc[0][["z", "y", "x"]])
        z0, y0, x0 = max(0, z - 3), max(0, y - 2), max(0, x - 2)
        np.testing.assert_allclose(spots["sum_prob_pixels"].iloc[0],
                                   prob[z0:z0 + 7, y0:y0 + 5, x0:x0 + 5].sum(), rtol=1e-6)
        np.testing.assert_allclose(spots["sum_decon_pixels"].iloc[0],
                                   want[z0:z0 + 7, y0:y0 + 5, x0:x0 + 5].sum(), rtol=1e-6)

    # resume: everything is on "disk", nothing is recomputed
    reg2 = dr_mod.DataRegistration(ds, decon_fiducial=True, decon_readout=True, verbose=1, feature_predictor=pred)
    reg2.register_all_tiles()
    assert pred.calls == 4 and "already have complete" in capsys.readouterr().out
    # overwrite_outputs: the bits are processed again
    reg2.overwrite_outputs = True
    reg2.perform_deformable_registration = False
    reg2._verbose = 0
    reg2.apply_registration_to_one_tile(0)
    assert pred.calls == 8


def test_no_deconvolution_paths_and_missing_predictor():
    dr_mod = pkg("DataRegistration")
    ds = _store()
    t = ds.tile_ids[0]
    pred = _Predictor()
    reg = dr_mod.DataRegistration(ds, decon_fiducial=False, decon_readout=False,
                                  perform_deformable_registration=False, verbose=0, feature_predictor=pred)
    reg.register_one_tile(t)
    for r, s in SHIFTS.items():
        T = ds.load_local_round_transform_zyx_um(t, r)
        np.testing.assert_allclose(T[:3, 3] / np.asarray(ds.voxel_size_zyx_um), s, atol=0.35)
    assert ds._fiducial_decon == {} and ds._readout_decon == {} and ds._flow == {}
    assert pred.calls == 4
    reg3 = dr_mod.DataRegistration(_store(), decon_fiducial=False, perform_deformable_registration=False, verbose=0)
    try:
        import ufish  # noqa: F401
    except ImportError:
        with pytest.raises(RuntimeError, match="feature predictor"):
            reg3.apply_registration_to_one_tile(0)
