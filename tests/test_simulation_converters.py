"""CPU: the statphysbio simulation converters and the F1 wrapper - file formats, the reference's
conventions (channel reversal, gain / offset, bit <-> round links, GT X/Y swap, z-centre shift) on known answers."""

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN, pkg


@pytest.fixture(scope="module")
def sim():
    return pkg("simulation")


def test_tiff_round_trip_and_shapes(sim, tmp_path):
    rng = np.random.default_rng(0)
    for dtype, shape in ((np.uint16, (3, 4, 10, 12)), (np.float32, (5, 7, 9)), (np.uint8, (6, 8))):
        a = (rng.uniform(0, 200, size=shape)).astype(dtype)
        sim.write_tiff(tmp_path / "a.tif", a)
        b = sim.read_tiff(tmp_path / "a.tif")
        assert b.dtype == a.dtype and b.shape == a.shape and np.array_equal(a, b)
    raw = (tmp_path / "a.tif").read_bytes()
    assert raw[:4] == b"II*\x00"          # classic little-endian TIFF, readable by any baseline reader


def test_metadata_csv_round_trip_matches_the_reference_format(sim, tmp_path):
    d = {"root_name": "data", "num_r": 8, "gain": 0.5, "blue_active": True, "flag": False, "name": "x y"}
    sim.write_metadata(d, tmp_path / "m.csv")
    text = (tmp_path / "m.csv").read_text().splitlines()
    assert text[0].startswith(",root_name,num_r") and text[1].startswith("0,data,8,0.5,True,False")
    back = sim.read_metadatafile(tmp_path / "m.csv")
    assert back["num_r"] == 8 and back["gain"] == 0.5 and back["blue_active"] is True and back["flag"] is False
    assert back["root_name"] == "data" and back[""] == 0


def test_synthetic_chromatic_affines_known_answer(sim):
    aff = sim.synthetic_chromatic_affines_zyx_um((16, 101, 201), (0.3, 0.1, 0.1), (0.58, 0.67, 0.52), shift_scale=2.0)
    assert sorted(aff) == [0.52, 0.58, 0.67]
    np.testing.assert_array_equal(aff[0.52], np.eye(4, dtype=np.float32))
    # 0.58: edge shift 0.5 * 2 = 1 px over a 100-px radius -> correction scale 1 / 1.01; z shift min(0.5, 1.25) px
    s = 1.0 / 1.01
    a = aff[0.58]
    assert a[1, 1] == pytest.approx(s, rel=1e-6) and a[2, 2] == pytest.approx(s, rel=1e-6)
    assert a[1, 3] == pytest.approx(5.0 * (1 - s), rel=1e-5) and a[2, 3] == pytest.approx(10.0 * (1 - s), rel=1e-5)
    assert a[0, 3] == pytest.approx(-0.5 * 0.3, rel=1e-6)
    assert aff[0.67][0, 3] == pytest.approx(-1.0 * 0.3, rel=1e-6)       # 0.5 * 2 = 1.0 <= cap 2.5


def _dataset(sim, tmp_path, **kw):
    root = sim.make_simulation(tmp_path / "sim", shape_zyx=(6, 32, 40), n_rna=40, n_beads=15, offset=100.0, gain=2.0,
                               seed=3, **kw)
    return root


def test_convert_simulation_writes_the_acquisition_layout(sim, tmp_path):
    root = _dataset(sim, tmp_path)
    flat = sim.read_tiff(root / "aligned_1.tiff")
    assert flat.shape == (8 * 3 * 6, 32, 40)
    acq = sim.convert_simulation(root)
    md = sim.read_metadatafile(acq / "scan_metadata.csv")
    assert md["num_r"] == 8 and md["num_ch"] == 3 and md["num_z"] == 6 and md["channels_reversed"] is True
    assert md["experiment_type"] == "3D" and md["gain"] == 2.0 and md["offset"] == 100.0
    for r in (1, 8):
        stem = f"data_r{r:04d}_tile0000"
        img = sim.read_tiff(acq / (stem + "_1") / (stem + ".tif"))
        assert img.shape == (3, 6, 32, 40) and img.dtype == np.uint16
        np.testing.assert_array_equal(img, flat.reshape(8, 3, 6, 32, 40)[r - 1])
        sp = sim.read_metadatafile(acq / (stem + "_stage_positions.csv"))
        # negative values stay strings (the reference's pattern has no sign, dataio.py:55-58): callers wrap in float()
        assert float(sp["stage_y"]) == pytest.approx(-0.098 * 16) and float(sp["stage_x"]) == pytest.approx(-0.098 * 20)
    assert (acq / "codebook.csv").exists() and (acq / "bit_order.csv").exists()


def test_convert_data_applies_gain_offset_channel_order_and_links(sim, tmp_path):
    root = _dataset(sim, tmp_path)
    acq = sim.convert_simulation(root)
    store = sim.convert_data(acq, z_step=2)
    flat = sim.read_tiff(root / "aligned_1.tiff").reshape(8, 3, 6, 32, 40)
    assert store.num_tiles == 1 and len(store.round_ids) == 8 and store.num_bits == 16
    assert store.voxel_size_zyx_um == pytest.approx((0.63, 0.098, 0.098)) and store.microscope_type == "3D"
    assert len(store.channel_psfs) == 3 and store.channel_psfs[0].shape == (3, 51, 51)
    assert all(abs(float(p.sum()) - 1.0) < 1e-5 for p in store.channel_psfs)
    assert store.datastore_state["Calibrations"] and store.datastore_state["Corrected"]
    t = store.tile_ids[0]

    def corrected(raw):   # (raw - offset) * gain, negatives to zero, truncation to uint16, every second plane
        v = (raw[::2].astype(np.float32) - 100.0) * 2.0
        v[v < 0] = 0
        return v.astype(np.uint16)

    for r in (0, 5):
        fid = store.load_local_corrected_image(t, round=store.round_ids[r], return_future=False)
        np.testing.assert_array_equal(fid, corrected(flat[r, 2]))          # blue is last in the file, first after the flip
        b1, b2 = 2 * r, 2 * r + 1                                          # bit_order: yellow then red
        np.testing.assert_array_equal(store.load_local_corrected_image(t, bit=store.bit_ids[b1], return_future=False),
                                      corrected(flat[r, 1]))
        np.testing.assert_array_equal(store.load_local_corrected_image(t, bit=store.bit_ids[b2], return_future=False),
                                      corrected(flat[r, 0]))
        assert store.load_local_round_linker(t, store.bit_ids[b1]) == r + 1
        assert store.load_local_wavelengths_um(t, bit=store.bit_ids[b1]) == pytest.approx((0.561, 0.580))
        assert store.load_local_wavelengths_um(t, bit=store.bit_ids[b2]) == pytest.approx((0.635, 0.670))
        assert store.load_local_wavelengths_um(t, round=store.round_ids[r]) == pytest.approx((0.488, 0.520))
    origin, c2s = store.load_local_stage_position_zyx_um(t, round=0)
    np.testing.assert_allclose(origin, [-1.57, -1.57, -1.96], atol=1e-6)   # z repeats stage_y (convert_to_datastore.py:383)
    np.testing.assert_array_equal(c2s, np.eye(4, dtype=np.float32))
    with pytest.raises(ValueError):
        sim.convert_data(acq, z_step=0)


def test_convert_data_sets_or_demands_the_round_links_of_a_factory_that_ignores_the_bit_order(sim, tmp_path):
    """a factory written to the older six-keyword contract drops ``experiment_order``; the converter
    then sets every bit's round link itself through ``set_round_linker`` - and refuses a store that offers no way to
    hold one instead of leaving every bit on round 1."""
    from conftest import pkg

    ds = pkg("datastore")
    acq = sim.convert_simulation(_dataset(sim, tmp_path))
    order = pd.read_csv(acq / "bit_order.csv").values

    def six_keyword_factory(**kw):
        keep = ("num_tiles", "num_rounds", "num_bits", "codebook", "voxel_size_zyx_um", "microscope_type")
        return ds.InMemoryDataStore(**{k: kw[k] for k in keep})

    store = sim.convert_data(acq, datastore_factory=six_keyword_factory)
    for r in range(order.shape[0]):
        for ch in (1, 2):
            assert store.load_local_round_linker(store.tile_ids[0], store.bit_ids[int(order[r, ch]) - 1]) == int(order[r, 0])

    class NoLinks(ds.InMemoryDataStore):
        set_round_linker = None

        def load_local_round_linker(self, tile, bit):
            return None

    def linkless_factory(**kw):
        keep = ("num_tiles", "num_rounds", "num_bits", "codebook", "voxel_size_zyx_um", "microscope_type")
        return NoLinks(**{k: kw[k] for k in keep})

    with pytest.raises(RuntimeError, match="round link"):
        sim.convert_data(acq, datastore_factory=linkless_factory)


def test_calculate_f1_conventions(sim, tmp_path):
    """A decoded table placed exactly on the ground truth scores 1; without the z-centre shift or with X and Y read
    the plain way it would not."""
    root = _dataset(sim, tmp_path)
    acq = sim.convert_simulation(root)
    store = sim.convert_data(acq)
    gt = pd.read_csv(root / "GT_spots.csv")
    genes = np.asarray(store.codebook["gene_id"].tolist())
    dz, dy, dx = store.voxel_size_zyx_um
    ny, nx = 32, 40
    # invert the reference's mapping: gt (Z, X, Y) -> decoded (global_z, global_y, global_x) with origin 0
    decoded = pd.DataFrame({"global_z": gt["Z"] - dz / 2, "global_y": gt["X"] + (ny / 2) * dy - dy / 2,
                            "global_x": gt["Y"] + (nx / 2) * dx - dx / 2, "gene_id": genes[gt["Gene_label"] - 1]})
    res = sim.calculate_F1(store, gt, search_radius=0.05, decoded_spots=decoded)
    assert res["F1 Score"] == pytest.approx(1.0) and res["False Positives"] == 0 and res["False Negatives"] == 0
    swapped = decoded.rename(columns={"global_y": "global_x", "global_x": "global_y"})
    assert sim.calculate_F1(store, gt, search_radius=0.05, decoded_spots=swapped)["F1 Score"] < 0.2
    unshifted = decoded.assign(global_z=decoded["global_z"] + dz / 2)
    assert sim.calculate_F1(store, gt, search_radius=0.05, decoded_spots=unshifted)["F1 Score"] < 0.2


def test_tiff_codec_against_the_real_tifffile(tmp_path):
    """tests/golden/tiff_codec.npz (tests/golden/make_golden_tiff.py, tifffile 2021.7.2 of the image's conda Python):
    ``read_tiff`` reads what tifffile wrote - including the RGB(A)-planar pages tifffile chooses on its own for a
    leading axis of length 3 or 4, which is the (channel, z, y, x) stack ``convert_simulation`` writes when z is 3 or
    4 - and ``write_tiff`` still produces, byte for byte, the files tifffile read back to the same arrays."""
    sim = pkg("simulation")
    g = np.load(GOLDEN / "tiff_codec.npz", allow_pickle=False)
    assert str(g["tifffile_version"]).startswith("2021.")
    for name in [str(c) for c in g["cases"]]:
        want = g["array_" + name]
        src = tmp_path / f"tifffile_{name}.tif"
        g["tifffile_file_" + name].tofile(src)
        got = sim.read_tiff(src)
        assert got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want), name
        dst = tmp_path / f"ours_{name}.tif"
        sim.write_tiff(dst, want)
        assert np.array_equal(np.fromfile(dst, dtype=np.uint8), g["ours_file_" + name]), name
        back = sim.read_tiff(dst)
        assert back.dtype == want.dtype and back.shape == want.shape and np.array_equal(back, want), name
