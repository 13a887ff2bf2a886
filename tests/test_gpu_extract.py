"""GPU parity: connected components + region statistics + the whole decode_one_tile."""

import numpy as np
import pandas as pd
import pytest
import torch

from conftest import pkg
from helpers import codebook_frame, sort_table, store_from_stack
from oracle import ccl_ref, decode_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("shape,n_codes,fill,is3d", [((6, 20, 24), 3, 0.5, True), ((1, 40, 40), 2, 0.6, True),
                                                     ((5, 17, 19), 4, 0.35, False), ((9, 9, 9), 1, 0.7, True)])
def test_ccl_partition_matches_skimage_semantics(shape, n_codes, fill, is3d):
    """Dense random label soups: every union-find race shows up as a split/merged region."""
    kernels = pkg("kernels")
    rng = np.random.default_rng(11)
    decoded = np.where(rng.uniform(size=shape) < fill, rng.integers(0, n_codes, size=shape), -1).astype(np.int16)
    zeros16 = torch.zeros(shape, dtype=torch.float16, device=DEV)
    table = kernels.connected_component_table(_t(decoded), torch.zeros((1, *shape), dtype=torch.float16, device=DEV),
                                              zeros16, zeros16, is_3d=is3d)
    roots = table["labels"].cpu().numpy().ravel()
    ref, n = ccl_ref.label_equal(decoded, is3d)
    assert len(table["root"]) == n
    rank = {int(r): i + 1 for i, r in enumerate(table["root"])}
    got = np.array([rank[int(r)] if r >= 0 else 0 for r in roots]).reshape(shape)
    assert np.array_equal(got, ref)
    areas = np.bincount(ref.ravel())[1:]
    assert np.array_equal(table["sums"][:, 0], areas.astype(np.float64))
    assert np.array_equal(table["code"], np.array([decoded.ravel()[r] for r in table["root"]]))


def test_region_table_matches_oracle_on_golden_decode(golden):
    kernels = pkg("kernels")
    g = golden("decode_pixels_22")
    table = kernels.connected_component_table(_t(g["decoded"]), _t(g["scaled"]), _t(g["magnitude"]),
                                              _t(g["distance"]), is_3d=True)
    labels, n = ccl_ref.label_equal(g["decoded"], True)
    ref = ccl_ref.region_table(labels, np.moveaxis(g["scaled"], 0, -1), g["magnitude"], g["distance"])
    assert len(ref) == n == len(table["root"])
    s = table["sums"]
    area = s[:, 0]
    np.testing.assert_array_equal(area, ref["area"].to_numpy())
    np.testing.assert_allclose(s[:, 1] / area, ref["centroid-0"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(s[:, 3] / area, ref["centroid-2"], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(table["min_distance"], ref["distance_min"].to_numpy(dtype=np.float32))
    B = g["scaled"].shape[0]
    means = (s[:, 11:].astype(np.float32) / area.astype(np.float32)[:, None]).astype(np.float16)
    refm = np.stack([ref[f"intensity_mean-{k}"].to_numpy() for k in range(B)], axis=1).astype(np.float16)
    diff = np.abs(means.astype(np.float32) - refm.astype(np.float32))
    assert np.mean(means.view(np.uint16) == refm.view(np.uint16)) > 0.999  # float16 double rounding is rare
    assert diff.max() <= 1e-3


def test_component_table_matches_scikit_image_golden(golden):
    """The HIP labelling and per-component sums against the real scikit-image 0.18.3
    (tests/golden/make_golden_skimage.py): labels bit for bit, float16 means to the double-rounding bound."""
    kernels = pkg("kernels")
    g, s = golden("decode_pixels_16"), golden("skimage_regionprops")
    for is3d, want in ((True, s["labels3"]), (False, s["labels2"])):
        table = kernels.connected_component_table(_t(g["decoded"]), _t(g["scaled"]), _t(g["magnitude"]),
                                                  _t(g["distance"]), is_3d=is3d)
        rank = {int(r): i + 1 for i, r in enumerate(table["root"])}
        roots = table["labels"].cpu().numpy().ravel()
        got = np.array([rank[int(r)] if r >= 0 else 0 for r in roots]).reshape(want.shape)
        assert np.array_equal(got, want)
        if not is3d:
            continue
        sums = table["sums"]
        keep = s["label"] - 1  # remove_small_objects(min_size=3) survivors, by scikit-image label
        area = sums[keep, 0]
        np.testing.assert_array_equal(area, s["area"])
        np.testing.assert_allclose(sums[keep, 1:4] / area[:, None], s["centroid"], rtol=0, atol=1e-12)
        np.testing.assert_array_equal(table["min_distance"][keep].astype(np.float32), s["distance_min"])
        means = (sums[keep, kernels.SUM_COLS:].astype(np.float32) / area.astype(np.float32)[:, None]).astype(np.float16)
        assert np.abs(means.astype(np.float64) - s["intensity_mean"]).max() <= 1e-3
        assert np.mean(means.astype(np.float64) == s["intensity_mean"]) > 0.99
        z, y, x = (sums[keep, 1 + k] / area for k in range(3))
        mzz, myy, mxx = sums[keep, 4] / area - z * z, sums[keep, 5] / area - y * y, sums[keep, 6] / area - x * x
        mzy, mzx, myx = sums[keep, 7] / area - z * y, sums[keep, 8] / area - z * x, sums[keep, 9] / area - y * x
        T = np.zeros((len(keep), 3, 3))
        T[:, 0, 0], T[:, 1, 1], T[:, 2, 2] = myy + mxx, mzz + mxx, mzz + myy
        T[:, 0, 1] = T[:, 1, 0] = -mzy
        T[:, 0, 2] = T[:, 2, 0] = -mzx
        T[:, 1, 2] = T[:, 2, 1] = -myx
        eig = np.clip(np.linalg.eigvalsh(T), 0, None)[:, ::-1]
        np.testing.assert_allclose(eig, s["inertia_tensor_eigvals"], rtol=1e-9, atol=1e-9)


def _decoder_for(golden_npz, lowpass, minimum_pixels=3):
    PD = pkg("PixelDecoder")
    g = golden_npz
    n_bits = g["codebook_bits"].shape[1]
    genes = list(g["gene_ids"]) if "gene_ids" in g.files else None
    cb = codebook_frame(g["codebook_bits"], genes)
    store = store_from_stack([g["lowpass"] if not lowpass else g["stack"]], cb)
    store.save_decode_normalization_vectors(None, "global", g["normalization"], g["background"])
    dec = PD.PixelDecoder(store, merfish_bits=n_bits, verbose=0,
                          excluded_gene_ids=[genes[i] for i in g["excluded_indices"]] if genes else None)
    return dec, cb


@pytest.mark.parametrize("name", ["decode_pixels_16", "decode_pixels_22"])
def test_decode_one_tile_matches_oracle_table(golden, name):
    """Same low-passed input on both sides: integers exact, floats to rounding."""
    g = golden(name)
    dec, cb = _decoder_for(g, lowpass=False)
    mag = tuple(g["magnitude_threshold"])
    res = dec.decode_one_tile(tile_idx=0, return_results=True, lowpass_sigma=None, magnitude_threshold=mag,
                              minimum_pixels=3, normalization_method="global")
    image, scaled, magnitude, distance, decoded = res
    assert decoded.dtype == np.int16 and scaled.dtype == np.float16 and image.dtype == np.float32
    assert np.array_equal(decoded, g["decoded"])                  # == the reference's own output
    matrix, genes, _pix, tr_thr, _blank = decode_ref.load_codebook(g["codebook_bits"], list(g["gene_ids"]))
    ref = ccl_ref.extract_barcodes(decoded, scaled, magnitude, distance, matrix, genes, tr_thr,
                                   minimum_pixels=3, spacing=(0.315, 0.098, 0.098))
    got = sort_table(dec.decoded_barcodes)
    ref = sort_table(ref)
    assert len(got) == len(ref) and len(got) > 20
    assert list(got.columns) == list(ref.columns)
    for col in ("gene_id", "barcode_id", "tile_z", "tile_y", "tile_x", "on_bit_1", "on_bit_4", "tile_idx"):
        assert (got[col].to_numpy() == ref[col].to_numpy()).all(), col
    np.testing.assert_array_equal(got["area"], ref["area"])
    for col in ("z", "y", "x", "global_z", "global_y", "global_x", "distance_min"):
        np.testing.assert_allclose(got[col].astype(float), ref[col].astype(float), rtol=0, atol=1e-6, err_msg=col)
    for col in ("inertia_tensor_eigvals-0", "inertia_tensor_eigvals-2"):
        np.testing.assert_allclose(got[col], ref[col], rtol=1e-9, atol=1e-9, err_msg=col)
    for col in ("bit01_mean_intensity", "bit07_mean_intensity", "magnitude_mean", "signal_mean", "s-b_mean"):
        np.testing.assert_allclose(got[col].astype(float), ref[col].astype(float), rtol=0, atol=1.5e-3, err_msg=col)
    assert (got["distance_min"] <= tr_thr).all()


def test_decode_one_tile_full_pipeline_with_lowpass(golden):
    """Low-pass included: the float32 filter differs from SciPy at 1e-7, calls agree > 99.9 %."""
    g = golden("decode_pixels_16")
    dec, _cb = _decoder_for(g, lowpass=True)
    res = dec.decode_one_tile(tile_idx=0, return_results=True, lowpass_sigma=(3.0, 1.0, 1.0),
                              magnitude_threshold=tuple(g["magnitude_threshold"]), minimum_pixels=3,
                              normalization_method="global")
    np.testing.assert_allclose(res[0], g["lowpass"], rtol=2e-6, atol=2e-4)
    agree = np.mean(res[4] == g["decoded"])
    assert agree > 0.999, agree
    truth = g["truth"]
    df = dec.decoded_barcodes
    found = 0
    for code, z, y, x in truth:
        d = np.abs(df["tile_z"] - z) + np.abs(df["tile_y"] - y) + np.abs(df["tile_x"] - x)
        found += int(((d <= 3) & (df["barcode_id"] == code + 1)).any())
    assert found >= 0.7 * len(truth)   # most planted emitters are recovered


def test_decode_one_tile_2d_mode_and_z_range(golden):
    g = golden("decode_pixels_2d_none")
    PD = pkg("PixelDecoder")
    cb = codebook_frame(g["codebook_bits"])
    store = store_from_stack([g["lowpass"]], cb, microscope_type="2D")
    dec = PD.PixelDecoder(store, merfish_bits=16, verbose=0)
    assert not dec._is_3D
    res = dec.decode_one_tile(return_results=True, lowpass_sigma=None, magnitude_threshold=(0.7, 10.0),
                              normalization_method="none")
    assert np.array_equal(res[4], g["decoded"])
    matrix, genes, _p, tr, _b = decode_ref.load_codebook(g["codebook_bits"], cb["gene_id"].tolist())
    ref = ccl_ref.extract_barcodes(res[4], res[1], res[2], res[3], matrix, genes, tr, is_3d=False,
                                   minimum_pixels=7, spacing=(0.315, 0.098, 0.098))
    assert len(dec.decoded_barcodes) == len(ref)
    dec3 = PD.PixelDecoder(store, merfish_bits=16, verbose=0, z_range=(1, 3), decode_mode="3d")
    out = dec3.decode_one_tile(return_results=True, lowpass_sigma=None, magnitude_threshold=(0.7, 10.0),
                               normalization_method="none")
    assert out[4].shape[0] == 2 and np.array_equal(out[4], g["decoded"][1:3])
    if len(dec3.decoded_barcodes):
        assert dec3.decoded_barcodes["z"].min() >= 1.0
    with pytest.raises(ValueError):
        PD.PixelDecoder(store, merfish_bits=16, verbose=0, z_range=(1, 2), decode_mode="3d").decode_one_tile(
            lowpass_sigma=None, normalization_method="none")


def test_decode_all_tiles_filters_and_saves_the_pooled_table(golden):
    """decode_all_tiles end to end on two overlapping copies of one tile: per-tile tables saved,
    pooled table blank-fraction filtered, cross-tile duplicates removed (PixelDecoder.py:4759-4870)."""
    from oracle import decode_ref  # noqa: F401  (helpers import)
    PD = pkg("PixelDecoder")
    post = pkg("postdecode")
    g = golden("decode_pixels_16")
    genes = list(g["gene_ids"])
    cb = codebook_frame(g["codebook_bits"], genes)
    store = store_from_stack([g["lowpass"], g["lowpass"]], cb)     # two tiles imaging the same field
    store.save_decode_normalization_vectors(None, "global", g["normalization"], g["background"])
    dec = PD.PixelDecoder(store, merfish_bits=g["codebook_bits"].shape[1], verbose=0)
    dec.decode_all_tiles(assign_to_cells=True, lowpass_sigma=None, magnitude_threshold=tuple(g["magnitude_threshold"]),
                         minimum_pixels=3, normalization_method="global", target_gross_misid_rate=0.05)
    per_tile = [store.load_local_decoded_spots(t) for t in store.tile_ids]
    assert len(per_tile[0]) > 20 and len(per_tile[0]) == len(per_tile[1])
    pooled = pd.concat(per_tile)
    pooled = pooled[pooled["gene_id"].notna()]
    want, _ = post.blank_fraction_filter(pooled, blank_count=dec._blank_count, barcode_count=dec._barcode_count,
                                         target_gross_misid_rate=0.05)
    want, _ = post.remove_duplicates_in_tile_overlap(want)
    saved = store.load_global_filtered_decoded_spots()
    assert list(saved.columns) == list(want.columns) and "cell_id" in saved.columns
    pd.testing.assert_frame_equal(saved.reset_index(drop=True), want.reset_index(drop=True))
    # identical tiles at identical stage positions: every transcript had a cross-tile twin, one survives
    assert 0 < len(saved) <= len(per_tile[0])
    assert dec._barcodes_filtered and (saved["cell_id"] == -1).all()
    # optimize_filtering re-runs the host stage from the saved per-tile tables
    dec2 = PD.PixelDecoder(store, merfish_bits=g["codebook_bits"].shape[1], verbose=0)
    dec2.optimize_filtering(assign_to_cells=False, target_gross_misid_rate=0.05)
    pd.testing.assert_frame_equal(store.load_global_filtered_decoded_spots().reset_index(drop=True),
                                  want.reset_index(drop=True))
