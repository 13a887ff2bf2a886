"""CPU: host logic of the DataRegistration mirror (DataRegistration.py:235-264, 1049-1083,
1346-1591, 2137-2141) - PSF choice, ROI sums, resume bookkeeping, work shares."""

import numpy as np
import pandas as pd
import pytest

from conftest import pkg


def _store(num_tiles=2, num_rounds=3, num_bits=4, shape=(4, 8, 8)):
    ds_mod = pkg("datastore")
    cb = pd.DataFrame({"gene_id": ["a"], **{f"bit{i + 1:02d}": [1] for i in range(num_bits)}})
    ds = ds_mod.InMemoryDataStore(num_tiles=num_tiles, num_rounds=num_rounds, num_bits=num_bits, codebook=cb)
    ds.channel_psfs = [np.ones((3, 3, 3), dtype=np.float32) * (i + 1) for i in range(3)]
    for t in ds.tile_ids:
        for r in ds.round_ids:
            ds.set_corrected_image(np.zeros(shape, dtype=np.uint16), tile=t, round=r)
        for b in ds.bit_ids:
            ds.set_corrected_image(np.zeros(shape, dtype=np.uint16), tile=t, bit=b)
    return ds


def test_resolve_psf_list_and_stack():
    dr = pkg("DataRegistration")
    psfs = [np.full((2, 3, 3), i, dtype=np.float64) for i in range(3)]
    assert dr._resolve_psf(psfs, 2).dtype == np.float32 and dr._resolve_psf(psfs, 2)[0, 0, 0] == 2
    assert dr._resolve_psf(np.stack(psfs), 1)[0, 0, 0] == 1
    with pytest.raises(IndexError):
        dr._resolve_psf(psfs, 3)
    with pytest.raises(IndexError):
        dr._resolve_psf(np.stack(psfs), -1)
    with pytest.raises(ValueError):
        dr._resolve_psf(np.zeros((3, 3)), 0)


def test_roi_sums_clip_like_the_reference():
    dr = pkg("DataRegistration")
    rng = np.random.default_rng(0)
    img = rng.uniform(size=(9, 12, 13)).astype(np.float32)
    spots = pd.DataFrame({"z": [0, 4, 8, 2], "y": [0, 6, 11, 1], "x": [0, 5, 12, 12]})
    got = dr._roi_sums(spots, img, (7, 5, 5))
    want = []
    for z, y, x in spots.to_numpy():
        z0, y0, x0 = max(0, z - 3), max(0, y - 2), max(0, x - 2)
        want.append(img[z0:min(9, z0 + 7), y0:min(12, y0 + 5), x0:min(13, x0 + 5)].sum())
    np.testing.assert_allclose(got, want, rtol=1e-6)


def test_constructor_defaults_properties_and_tile_setter(capsys):
    dr = pkg("DataRegistration")
    sf = pkg("utils.sofima_registration")
    ds = _store()
    reg = dr.DataRegistration(ds, verbose=0)
    assert reg._decon_fiducial and not reg._decon_readout and reg._perform_deformable_registration
    assert reg._crop_yx_decon == 2048 and reg._num_gpus == 1 and not reg._global_registration
    assert isinstance(reg._sofima_config, sf.SofimaRegistrationConfig)
    assert reg.tile_id is None and "not defined" in capsys.readouterr().out
    reg.tile_id = 1
    assert reg.tile_id == "tile0001"
    reg.tile_id = "tile0000"
    assert reg.tile_id == "tile0000"
    reg.tile_id = "nope"
    assert reg.tile_id == "tile0000" and "valid tile id" in capsys.readouterr().out
    reg.overwrite_outputs = True
    reg.perform_deformable_registration = False
    assert reg.overwrite_outputs and not reg.perform_deformable_registration
    with pytest.raises(NotImplementedError):
        reg.global_register()


def test_resume_bookkeeping():
    dr = pkg("DataRegistration")
    ds = _store()
    reg = dr.DataRegistration(ds, verbose=0)
    t0 = ds.tile_ids[0]
    assert not reg._is_tile_complete(t0)
    for r in ds.round_ids:
        ds.save_local_deconvolved_fiducial_image(np.zeros((4, 8, 8), dtype=np.uint16), tile=t0, round=r)
        ds.save_local_round_transform_zyx_um(np.eye(4), tile=t0, round=r)
    assert reg._has_valid_deconvolved_fiducial_image(tile_id=t0, round_id=ds.round_ids[1])
    assert not reg._is_tile_complete(t0)           # feature predictor outputs missing
    for b in ds.bit_ids:
        ds.save_local_feature_predictor_image(np.zeros((4, 8, 8), dtype=np.float32), tile=t0, bit=b)
    assert not reg._has_valid_feature_predictor_outputs(tile_id=t0, bit_id=ds.bit_ids[0])   # spots missing
    for b in ds.bit_ids:
        ds.save_local_feature_predictor_spots(pd.DataFrame({"z": []}), tile=t0, bit=b)
    assert reg._is_tile_complete(t0) and not reg._is_tile_complete(ds.tile_ids[1])
    # a decon volume of the wrong shape is not valid
    ds.save_local_deconvolved_fiducial_image(np.zeros((4, 8, 7), dtype=np.uint16), tile=t0, round=ds.round_ids[0])
    assert not reg._is_tile_complete(t0)
    # without fiducial deconvolution the decon volumes are not required
    assert dr.DataRegistration(ds, decon_fiducial=False, verbose=0)._is_tile_complete(t0)
    assert not reg._has_valid_deconvolved_readout_image(tile_id=t0, bit_id=ds.bit_ids[0])
    ds.save_local_deconvolved_readout_image(np.zeros((4, 8, 8), dtype=np.uint16), tile=t0, bit=ds.bit_ids[0])
    assert reg._has_valid_deconvolved_readout_image(tile_id=t0, bit_id=ds.bit_ids[0])
    with pytest.raises(ValueError):
        reg._has_valid_deconvolved_fiducial_image(tile_id=t0)
    with pytest.raises(ValueError):
        reg._shape_of(t0, "corrected_data")


def test_shares_are_contiguous_ceil_chunks():
    dr = pkg("DataRegistration")
    ds = _store()
    with pytest.raises(RuntimeError, match="in-memory"):
        dr.DataRegistration(ds, num_gpus=2, verbose=0)._shares(["a", "b"])
    ds.in_memory = False   # stands for a path-backed store
    items = [f"r{i}" for i in range(7)]
    for gpus, want in [(1, [items]), (2, [items[:4], items[4:]]), (3, [items[:3], items[3:6], items[6:]]),
                       (16, [[i] for i in items])]:
        shares, distributed = dr.DataRegistration(ds, num_gpus=gpus, verbose=0)._shares(items)
        assert shares == want and not distributed
    assert dr.DataRegistration(ds, verbose=0)._shares([]) == ([], False)


def test_no_gpus_is_an_error():
    dr = pkg("DataRegistration")
    reg = dr.DataRegistration(_store(), num_gpus=0, verbose=0)
    reg.tile_id = 0
    with pytest.raises(RuntimeError):
        reg._apply_registration_to_bits()


def test_audit_flags_the_round_that_sits_an_image_width_off():
    """Round 4: a wrapped twin of the true shift (phase_cross_correlation's three-sample corner tile) is made visible -
    the shifts of the F1 fixture's mis-registered dataset: seven rounds within a few pixels, one 254 px away."""
    dr = pkg("DataRegistration")
    ds = _store(num_tiles=1, num_rounds=8, shape=(16, 256, 256))
    spacing = np.asarray(ds.voxel_size_zyx_um, dtype=np.float64)
    found = [[0.0, 0.0, 0.0], [0.3, 2.4, -2.5], [-0.2, -1.1, 3.8], [0.1, 3.9, 1.3], [-0.0, -2.7, -4.1],
             [5.9, -254.2, 125.6], [-0.3, -3.5, -1.5], [0.1, 2.5, -2.3]]
    t = ds.tile_ids[0]
    for rid, shift in zip(ds.round_ids, found):
        T = np.eye(4, dtype=np.float32)
        T[:3, 3] = np.asarray(shift) * spacing
        ds.save_local_round_transform_zyx_um(T, tile=t, round=rid)
    reg = dr.DataRegistration(ds, verbose=0)
    reg.tile_id = t
    suspects = reg._audit_round_transforms()
    assert [s[0] for s in suspects] == [ds.round_ids[5]]
    np.testing.assert_allclose(suspects[0][1], found[5], atol=0.02)      # shifts are held as float32 microns
    assert reg.suspect_registrations == {t: suspects}
    # the same tile with that round repaired: nothing to report, and the record is cleared
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = np.asarray([0.5, 1.2, 3.9]) * spacing
    ds.save_local_round_transform_zyx_um(T, tile=t, round=ds.round_ids[5])
    assert reg._audit_round_transforms() == [] and reg.suspect_registrations == {}
