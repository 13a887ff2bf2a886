"""GPU: ``optimize_normalization_by_decoding`` and the decode that follows it against a run of the REFERENCE's own
driver (PixelDecoder.py:4581-4757, worker :312-414, decode_one_tile :4471-4579; tests/golden/make_golden.py::
make_optimize): global vectors, the vectors after each of three iterations, the pooled transcript count per iteration
and the final tile's transcript table."""

import numpy as np
import pandas as pd
import pytest

from conftest import pkg
from helpers import codebook_frame, sort_table, use_scipy_lowpass

pytestmark = pytest.mark.gpu
KW = dict(lowpass_sigma=(3.0, 1.0, 1.0), magnitude_threshold=(0.9, 10.0), minimum_pixels=3)


def _store(g, to_device=None):
    ds_mod = pkg("datastore")
    cb = codebook_frame(g["codebook_bits"], [str(x) for x in g["gene_ids"]])
    n_tiles, n_bits = g["readouts"].shape[:2]
    store = ds_mod.InMemoryDataStore(num_tiles=n_tiles, num_rounds=1, num_bits=n_bits, codebook=cb,
                                     voxel_size_zyx_um=(0.315, 0.098, 0.098))
    for t in range(n_tiles):
        for b in range(n_bits):
            store.set_corrected_image(g["readouts"][t, b], tile=t, bit=b)
            store.save_local_feature_predictor_image(g["predictors"][t, b], tile=t, bit=b)
        store.set_stage_position_zyx_um(store.tile_ids[t], np.array([10.0, 50.0 * t, 0.0], dtype=np.float32))
    return store, cb


def _run_optimize(PD, g):
    store, _cb = _store(g)
    history, counts = [], []
    save = store.save_decode_normalization_vectors

    def save_rec(decode_run_key, kind, normalization, background, *a, **k):
        history.append((kind, np.asarray(normalization, dtype=np.float32).copy(),
                        np.asarray(background, dtype=np.float32).copy()))
        return save(decode_run_key, kind, normalization, background, *a, **k)

    store.save_decode_normalization_vectors = save_rec
    dec = PD.PixelDecoder(store, merfish_bits=g["codebook_bits"].shape[1], num_gpus=1, verbose=0)
    load_all = dec._load_all_barcodes

    def load_all_rec(*a, **k):
        load_all(*a, **k)
        counts.append(len(dec._df_barcodes_loaded))

    dec._load_all_barcodes = load_all_rec
    dec.optimize_normalization_by_decoding(n_iterations=3, tile_indices=[0, 1], **KW)
    store.save_decode_normalization_vectors = save
    ref_kinds = [str(k) for k in g["kinds"]]
    assert history[0][0] == "global" and ref_kinds[0] == "global"
    # one entry per iteration on each side (the reference stores every result twice)
    mine = [h for h in history if h[0] == "iterative"]
    ref_n = [g["normalization_history"][i] for i, k in enumerate(ref_kinds) if k == "iterative"][::2]
    ref_b = [g["background_history"][i] for i, k in enumerate(ref_kinds) if k == "iterative"][::2]
    mine = mine[:: max(1, len(mine) // 3)][:3]
    assert len(mine) == 3 and len(ref_n) == 3 and len(counts) == 3
    return store, save, history, mine, ref_n, ref_b, counts


def test_optimize_normalization_is_exact_behind_the_reference_runs_low_pass(golden, monkeypatch):
    """With the low-pass the reference run used (SciPy, float64 accumulation) in front of the HIP decode / extract /
    update loop, nothing is left that may differ: the global vectors, the vectors after each of the three iterations,
    the pooled transcript count per iteration, the decoded volume of the final tile and its table are the
    reference's - integers and rounded medians exactly (north_star: integer spot indices bit-exact)."""
    PD = pkg("PixelDecoder")
    g = golden("optimize_normalization")
    use_scipy_lowpass(monkeypatch)
    store, save, history, mine, ref_n, ref_b, counts = _run_optimize(PD, g)
    # exact order statistics of the same float32 values
    np.testing.assert_array_equal(history[0][1], g["normalization_history"][0])
    np.testing.assert_array_equal(history[0][2], g["background_history"][0])
    assert list(counts) == list(g["barcode_counts"])
    for it in range(3):
        np.testing.assert_array_equal(mine[it][1], ref_n[it], err_msg=f"normalisation, iteration {it}")
        np.testing.assert_array_equal(mine[it][2], ref_b[it], err_msg=f"background, iteration {it}")
    dec2 = PD.PixelDecoder(store, merfish_bits=g["codebook_bits"].shape[1], num_gpus=1, verbose=0)
    res = dec2.decode_one_tile(tile_idx=1, return_results=True, normalization_method="iterative", **KW)
    assert np.array_equal(res[4], g["decoded"])
    want = sort_table(pd.DataFrame({str(c): g["col_" + str(c)] for c in g["table_columns"]}))
    got = sort_table(dec2.decoded_barcodes)
    assert list(got.columns) == list(want.columns) and len(got) == len(want)
    assert (got["gene_id"].to_numpy().astype(str) == want["gene_id"].to_numpy().astype(str)).all()
    for c in ("area", "barcode_id", "tile_idx", "tile_z", "tile_y", "tile_x"):
        np.testing.assert_array_equal(got[c].to_numpy().astype(np.int64), want[c].to_numpy().astype(np.int64), err_msg=c)
    on = lambda tb: np.sort(tb[["on_bit_1", "on_bit_2", "on_bit_3", "on_bit_4"]].to_numpy(), axis=1)  # noqa: E731
    np.testing.assert_array_equal(on(got), on(want))
    for c in ("global_z", "global_y", "global_x"):   # rounded to 2 decimals: equal, or one float32-rounding step apart
        np.testing.assert_allclose(got[c].astype(float), want[c].astype(float), rtol=0, atol=0.0101, err_msg=c)
    for c in ("distance_min", "signal_mean", "bkd_mean", "magnitude_mean"):   # float16 volumes: <= one ulp
        np.testing.assert_allclose(got[c].astype(float), want[c].astype(float), rtol=0, atol=1.5e-3, err_msg=c)


def test_optimize_normalization_with_the_hip_low_pass_tracks_the_reference_run(golden):
    """The same loop with the production float32 low-pass (differs from SciPy's float64 accumulation by ~1e-6
    relative): medians rounded to 0.1 may move by one rounding step where they sat on a tie, a transcript whose voxel
    sat on a threshold may come or go."""
    PD = pkg("PixelDecoder")
    g = golden("optimize_normalization")
    store, save, history, mine, ref_n, ref_b, counts = _run_optimize(PD, g)
    np.testing.assert_allclose(history[0][1], g["normalization_history"][0], rtol=2e-5)
    np.testing.assert_allclose(history[0][2], g["background_history"][0], rtol=2e-5, atol=1e-5)
    for it in range(3):
        dn, db = np.abs(mine[it][1] - ref_n[it]), np.abs(mine[it][2] - ref_b[it])
        assert dn.max() <= 0.1001 and db.max() <= 0.1001, (it, dn.max(), db.max())
        assert np.mean(dn < 1e-4) >= 0.8 and np.mean(db < 1e-4) >= 0.8, (it, mine[it][1], ref_n[it])
    assert np.all(np.abs(np.asarray(counts) - g["barcode_counts"]) <= 2), (counts, g["barcode_counts"])
    # the decode that follows, on the reference's own final vectors so that a 0.1 step above cannot leak into it
    save(None, "iterative", ref_n[-1], ref_b[-1])
    save(None, "global", g["normalization_history"][0], g["background_history"][0])
    dec2 = PD.PixelDecoder(store, merfish_bits=g["codebook_bits"].shape[1], num_gpus=1, verbose=0)
    res = dec2.decode_one_tile(tile_idx=1, return_results=True, normalization_method="iterative", **KW)
    agree = np.mean(res[4] == g["decoded"])
    assert agree > 0.9995, agree
    want = sort_table(pd.DataFrame({str(c): g["col_" + str(c)] for c in g["table_columns"]}))
    got = sort_table(dec2.decoded_barcodes)
    assert list(got.columns) == list(want.columns)
    assert abs(len(got) - len(want)) <= 2
    key = ["tile_z", "tile_y", "tile_x", "gene_id"]
    m = got.assign(gene_id=got["gene_id"].astype(str)).merge(want.assign(gene_id=want["gene_id"].astype(str)), on=key,
                                                             suffixes=("", "_ref"))
    assert len(m) >= len(want) - 2
    assert np.mean(m["area"] == m["area_ref"]) > 0.97
    np.testing.assert_allclose(m["global_y"], m["global_y_ref"], atol=0.0101)
    np.testing.assert_allclose(m["distance_min"].astype(float), m["distance_min_ref"].astype(float), atol=2e-3)
    np.testing.assert_allclose(m["signal_mean"].astype(float), m["signal_mean_ref"].astype(float), atol=2e-3)


def test_optimize_normalization_is_the_same_from_hbm_resident_inputs(golden):
    """Inputs already in HBM (the bench's configuration) give the same vectors as host arrays."""
    import torch

    PD = pkg("PixelDecoder")
    g = golden("optimize_normalization")
    out = []
    for on_device in (False, True):
        store, _ = _store(g)
        if on_device:
            for k, v in list(store._corrected.items()):
                store._corrected[k] = torch.from_numpy(np.ascontiguousarray(v)).to("cuda:0")
            for k, v in list(store._predictor.items()):
                store._predictor[k] = torch.from_numpy(np.ascontiguousarray(v)).to("cuda:0")
        dec = PD.PixelDecoder(store, merfish_bits=g["codebook_bits"].shape[1], num_gpus=1, verbose=0)
        dec.optimize_normalization_by_decoding(n_iterations=2, tile_indices=[0, 1], **KW)
        out.append(store.load_decode_normalization_vectors(None, "iterative"))
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
