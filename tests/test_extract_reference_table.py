"""The transcript table of ``PixelDecoder._extract_barcodes`` against a run of the REFERENCE's own function
(PixelDecoder.py:2908-3201, tests/golden/make_golden.py::make_extract: the reference's table assembly executed over
scikit-image-pinned label / regionprops stand-ins) - CPU: the oracle restatement, GPU: the HIP path."""

import numpy as np
import pandas as pd
import pytest

from conftest import pkg
from helpers import codebook_frame, sort_table, store_from_stack
from oracle import ccl_ref, decode_ref

CASES = ["extract_barcodes_16", "extract_barcodes_22", "extract_barcodes_2d"]


def _fixture_table(g) -> pd.DataFrame:
    return pd.DataFrame({str(c): g["col_" + str(c)] for c in g["columns"]})


def _src(golden, g):
    return golden(str(g["source"]))


@pytest.mark.parametrize("name", CASES)
def test_oracle_table_equals_the_reference_run(golden, name):
    g = golden(name)
    d = _src(golden, g)
    want = _fixture_table(g)
    matrix, genes, _pix, tr_thr, _blank = decode_ref.load_codebook(d["codebook_bits"], [str(x) for x in d["gene_ids"]])
    got = ccl_ref.extract_barcodes(
        d["decoded"], d["scaled"], d["magnitude"], d["distance"], matrix, genes, tr_thr, is_3d=bool(g["is_3d"]),
        minimum_pixels=int(g["minimum_pixels"]), tile_idx=int(g["tile_idx"]), spacing=g["spacing"], origin=g["origin"],
        affine=g["affine"], camera_to_stage=g["camera_to_stage"])
    assert list(got.columns) == list(want.columns)
    assert len(got) == len(want) and len(want) > 20
    for c in want.columns:
        a, b = got[c].to_numpy(), want[c].to_numpy()
        if c.startswith("on_bit_"):
            continue  # argsort of a boolean row without kind=: a set for B > 16; compared below
        if a.dtype == object or b.dtype.kind in "US":
            assert (a.astype(str) == b.astype(str)).all(), c
        else:
            np.testing.assert_array_equal(a.astype(np.float64), b.astype(np.float64), err_msg=c)
    on = lambda t: np.sort(t[["on_bit_1", "on_bit_2", "on_bit_3", "on_bit_4"]].to_numpy(), axis=1)  # noqa: E731
    np.testing.assert_array_equal(on(got), on(want))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_table_equals_the_reference_run(golden, name):
    """decode_one_tile from the fixture's low-passed stack: the decoded volume is the reference's bit for bit, so the
    HIP component table must reproduce the reference's transcript table - integers and coordinates exactly, float16
    derived means to the double-rounding bound."""
    PD = pkg("PixelDecoder")
    g = golden(name)
    d = _src(golden, g)
    want = sort_table(_fixture_table(g))
    is_3d, tile_idx = bool(g["is_3d"]), int(g["tile_idx"])
    genes = [str(x) for x in d["gene_ids"]]
    cb = codebook_frame(d["codebook_bits"], genes)
    store = store_from_stack([d["lowpass"]] * (tile_idx + 1), cb, microscope_type="3D" if is_3d else "2D",
                             voxel=tuple(float(v) for v in g["spacing"]))
    store.save_decode_normalization_vectors(None, "global", d["normalization"], d["background"])
    t = store.tile_ids[tile_idx]
    store.set_stage_position_zyx_um(t, g["origin"], g["camera_to_stage"])
    store.set_global_coord_xforms_um(t, g["affine"], g["origin"], g["spacing"])
    dec = PD.PixelDecoder(store, merfish_bits=d["codebook_bits"].shape[1], verbose=0,
                          excluded_gene_ids=[genes[i] for i in d["excluded_indices"]])
    res = dec.decode_one_tile(tile_idx=tile_idx, return_results=True, lowpass_sigma=None,
                              magnitude_threshold=tuple(d["magnitude_threshold"]),
                              minimum_pixels=int(g["minimum_pixels"]), normalization_method="global")
    assert np.array_equal(res[4], d["decoded"])
    got = sort_table(dec.decoded_barcodes)
    assert list(got.columns) == list(want.columns)
    assert len(got) == len(want)
    for c in ("gene_id",):
        assert (got[c].to_numpy().astype(str) == want[c].to_numpy().astype(str)).all(), c
    for c in ("area", "barcode_id", "tile_idx", "tile_z", "tile_y", "tile_x"):
        np.testing.assert_array_equal(got[c].to_numpy().astype(np.int64), want[c].to_numpy().astype(np.int64), err_msg=c)
    for c in ("z", "y", "x"):
        np.testing.assert_allclose(got[c].astype(float), want[c].astype(float), rtol=0, atol=1e-9, err_msg=c)
    for c in ("global_z", "global_y", "global_x"):   # rounded to 2 decimals: equal, or one float32-rounding step apart
        np.testing.assert_allclose(got[c].astype(float), want[c].astype(float), rtol=0, atol=0.0101, err_msg=c)
        assert np.mean(got[c].to_numpy() == want[c].to_numpy()) > 0.95, c
    for c in ("inertia_tensor_eigvals-0", "inertia_tensor_eigvals-1", "inertia_tensor_eigvals-2"):
        np.testing.assert_allclose(got[c], want[c], rtol=1e-9, atol=1e-9, err_msg=c)
    n_bits = d["codebook_bits"].shape[1]
    f16_cols = [f"bit{i:02d}_mean_intensity" for i in range(1, n_bits + 1)] + ["magnitude_mean", "distance_min",
                                                                              "signal_mean", "bkd_mean", "s-b_mean"]
    for c in f16_cols:   # float16 volumes differ from the reference's in <= 0.1 % of voxels by one ulp
        np.testing.assert_allclose(got[c].astype(float), want[c].astype(float), rtol=0, atol=1.5e-3, err_msg=c)
    on = lambda tb: np.sort(tb[["on_bit_1", "on_bit_2", "on_bit_3", "on_bit_4"]].to_numpy(), axis=1)  # noqa: E731
    np.testing.assert_array_equal(on(got), on(want))
