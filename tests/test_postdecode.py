"""CPU: post-decode filtering, duplicate removal (PixelDecoder.py:3386-3846, 4137-4363) and the F1
scorer (cli/statphysbio_simulation/calculate_F1.py:19-268) against outputs of the reference's own
functions on a synthetic transcript table (tests/golden/postdecode.npz, made by make_golden.py)."""

import numpy as np
import pandas as pd
import pytest

from conftest import pkg


def _table(g, rows=None):
    df = pd.DataFrame({"gene_id": g["gene_sorted"][g["gene_code"]], "magnitude_mean": g["magnitude_mean"],
                       "area": g["area"], "distance_min": g["distance_min"], "tile_idx": g["tile_idx"],
                       "global_z": g["global_z"], "global_y": g["global_y"], "global_x": g["global_x"]})
    return df if rows is None else df.loc[rows]


def _check(g, tag, filtered, diag):
    np.testing.assert_array_equal(filtered.index.to_numpy(), g[f"{tag}_kept_rows"])
    np.testing.assert_array_equal(filtered["blank_fraction"].to_numpy(), g[f"{tag}_kept_fraction"])
    np.testing.assert_array_equal(filtered["blank_fraction_bin"].to_numpy(), g[f"{tag}_kept_bin"])
    assert (filtered["cell_id"] == -1).all()
    for k in ("all_histogram", "blank_histogram", "blank_fraction_histogram", "intensity_bins", "voxel_number_bins",
              "vector_distance_bins"):
        np.testing.assert_array_equal(np.asarray(diag[k]), g[f"{tag}_{k}"], err_msg=k)
    np.testing.assert_array_equal(np.asarray(diag["chosen_threshold"]), g[f"{tag}_chosen_threshold"])
    np.testing.assert_array_equal(np.asarray(diag["achieved_gross_misid_rate"]), g[f"{tag}_achieved_gross_misid_rate"])
    assert bool(diag["target_reached"]) == bool(g[f"{tag}_target_reached"])
    np.testing.assert_array_equal(diag["threshold_sweep"].to_numpy(dtype=float), g[f"{tag}_sweep"])
    assert diag.get("reason", "") == str(g[f"{tag}_reason"])


def test_blank_fraction_filter_matches_reference_run(golden):
    pd_mod = pkg("postdecode")
    g = golden("postdecode")
    table = _table(g)
    kw = dict(blank_count=10, barcode_count=140)
    _check(g, "auto", *pd_mod.blank_fraction_filter(table, target_gross_misid_rate=0.05, **kw))
    _check(g, "strict", *pd_mod.blank_fraction_filter(table, target_gross_misid_rate=1e-4, **kw))
    _check(g, "explicit", *pd_mod.blank_fraction_filter(
        table, target_gross_misid_rate=0.1, intensity_bins=[0.0, 0.8, 1.5, 2.5, 10.0],
        voxel_number_bins=[15.5, 20.5, 40.5, 100.5], vector_distance_bins=np.linspace(0.0, 0.52, 6), **kw))
    few = table.iloc[:40].copy()
    few["area"] = g["few_area"]
    _check(g, "few", *pd_mod.blank_fraction_filter(few, target_gross_misid_rate=0.05, **kw))
    _check(g, "noblank", *pd_mod.blank_fraction_filter(_table(g, g["noblank_rows"]), target_gross_misid_rate=0.05, **kw))


def test_blank_fraction_filter_edge_cases():
    pd_mod = pkg("postdecode")
    cols = {"gene_id": pd.Series(dtype="string"), "magnitude_mean": pd.Series(dtype=np.float32),
            "area": pd.Series(dtype=np.float32), "distance_min": pd.Series(dtype=np.float32)}
    out, diag = pd_mod.blank_fraction_filter(pd.DataFrame(cols), blank_count=10, barcode_count=140)
    assert out.empty and diag["reason"] == "no_transcripts" and "cell_id" in out.columns
    df = pd.DataFrame({"gene_id": ["a", "Blank-1"], "magnitude_mean": [np.nan, np.inf], "area": [20.0, 20.0],
                       "distance_min": [0.1, 0.2]})
    out, diag = pd_mod.blank_fraction_filter(df, blank_count=10, barcode_count=140)
    assert out.empty and diag["reason"] == "no_valid_features"
    df["magnitude_mean"] = [1.0, 2.0]
    out, diag = pd_mod.blank_fraction_filter(df, blank_count=0, barcode_count=140)
    assert len(out) == 2 and diag["reason"] == "no_blank_barcodes"
    with pytest.raises(ValueError, match="requires columns"):
        pd_mod.blank_fraction_filter(df.drop(columns=["area"]), blank_count=1, barcode_count=2)
    with pytest.raises(ValueError, match="at least two finite"):
        pd_mod.blank_fraction_filter(df, blank_count=1, barcode_count=2, intensity_bins=[1.0, np.nan])
    # one transcript per class, identical features: a single bin with blank fraction 0.5
    df2 = pd.DataFrame({"gene_id": ["a", "Blank-1"], "magnitude_mean": [1.0, 1.0], "area": [20.0, 20.0],
                        "distance_min": [0.1, 0.1]})
    out, diag = pd_mod.blank_fraction_filter(df2, blank_count=10, barcode_count=140, target_gross_misid_rate=0.05)
    assert not diag["target_reached"] and diag["chosen_threshold"] == 0.5 and len(out) == 2
    assert diag["achieved_gross_misid_rate"] == (1 / 10) / (2 / 140)


def test_duplicate_removal_matches_reference_run(golden):
    pd_mod = pkg("postdecode")
    g = golden("postdecode")
    kept = _table(g, g["auto_kept_rows"]).copy()
    kept["row_id"] = kept.index.to_numpy()
    within, dropped = pd_mod.remove_duplicates_within_tile(kept, radius_xy=0.098, radius_z=0.31)
    np.testing.assert_array_equal(within["row_id"].to_numpy(), g["within_rows"])
    assert len(dropped) == len(kept) - len(within) > 0
    overlap, dmin = pd_mod.remove_duplicates_in_tile_overlap(within, radius=0.75)
    np.testing.assert_array_equal(overlap["row_id"].to_numpy(), g["overlap_rows"])
    assert len(within) - len(overlap) > 0 and list(overlap.index) == list(range(len(overlap)))
    # nothing to do
    one = kept.iloc[:1]
    assert pd_mod.remove_duplicates_within_tile(one)[0] is one


def test_duplicate_rules_on_hand_made_cases():
    pd_mod = pkg("postdecode")
    df = pd.DataFrame({
        "gene_id": ["a", "a", "a", "b", "a", "a"],
        "tile_idx": [0, 0, 0, 0, 1, 1],
        "global_z": [0.0, 0.3, 0.6, 0.3, 0.0, 0.0],
        "global_y": [10.0, 10.01, 10.02, 10.0, 10.0, 50.0],
        "global_x": [5.0, 5.0, 5.0, 5.0, 5.0, 5.0],
        "distance_min": [0.3, 0.1, 0.2, 0.05, 0.3, 0.4],
    })
    out, dropped = pd_mod.remove_duplicates_within_tile(df, radius_xy=0.1, radius_z=0.3)
    # rows 0-1-2 chain across planes (same gene) -> keep row 1; row 3 other gene; tile 1 untouched
    assert dropped == [0, 2] and list(out["distance_min"]) == [0.1, 0.05, 0.3, 0.4]
    out2, d2 = pd_mod.remove_duplicates_in_tile_overlap(df, radius=0.05)
    # rows 0 (tile 0) and 4 (tile 1) coincide with equal distance_min: the later row is dropped
    assert len(out2) == 5 and list(d2) == [0.3] and 0.4 in list(out2["distance_min"])
    same_plane = df.copy()
    same_plane["global_z"] = 0.0
    assert pd_mod.remove_duplicates_within_tile(same_plane, radius_xy=0.1, radius_z=0.3)[1] == []


def test_f1_scorer_matches_reference_run(golden):
    sc = pkg("scoring")
    pd_mod = pkg("postdecode")
    g = golden("postdecode")
    final = _table(g).loc[g["overlap_rows"]]
    q, qg = final[["global_z", "global_y", "global_x"]].to_numpy(), final["gene_id"].to_numpy()
    for tag, radius in (("r1", 1.0), ("r03", 0.3)):
        res, tp, fp, fn = sc.calculate_F1_with_radius(q, qg, g["f1_gt"], g["f1_gt_gene"], radius)
        got = [res["F1 Score"], res["Precision"], res["Recall"], res["True Positives"], res["False Positives"],
               res["False Negatives"]]
        np.testing.assert_array_equal(np.asarray(got, dtype=float), g[f"f1_{tag}_metrics"])
        np.testing.assert_array_equal(tp, g[f"f1_{tag}_tp"])
        np.testing.assert_array_equal(fp, g[f"f1_{tag}_fp"])
        np.testing.assert_array_equal(fn, g[f"f1_{tag}_fn"])
    assert pd_mod  # imported for symmetry with the other tests


def test_f1_scorer_trivial_cases_and_coordinate_conventions():
    sc = pkg("scoring")
    e = np.empty((0, 3))
    assert sc.calculate_F1_with_radius(e, [], e, [], 1.0)[0]["F1 Score"] == 1.0
    r = sc.calculate_F1_with_radius(e, [], [[0, 0, 0]], ["a"], 1.0)
    assert r[0]["False Negatives"] == 1 and r[0]["F1 Score"] == 0.0 and r[3].shape == (1, 3)
    r = sc.calculate_F1_with_radius([[0, 0, 0]], ["a"], e, [], 1.0)
    assert r[0]["False Positives"] == 1 and r[2].shape == (1, 3)
    # same place, different gene: never matched; one-to-one: two decoded near one truth -> one TP
    r = sc.calculate_F1_with_radius([[0, 0, 0], [0, 0, 0.1], [5, 5, 5]], ["a", "a", "b"],
                                    [[0, 0, 0.04], [5, 5, 5]], ["a", "c"], 0.5)[0]
    assert (r["True Positives"], r["False Positives"], r["False Negatives"]) == (1, 2, 1)
    # score_against_ground_truth: decoded z -> voxel centre, GT columns (Z, X, Y) swapped and re-centred
    voxel, shape = (0.3, 0.1, 0.1), (10, 64, 64)
    decoded = pd.DataFrame({"global_z": [0.9, 0.0], "global_y": [1.0, 2.0], "global_x": [3.0, 4.0],
                            "gene_id": ["g2", "g1"]})
    off = 64 / 2 * 0.1 - 0.05
    gt = pd.DataFrame({"Z": [1.05, 0.01], "X": [1.0 - off, 2.0 - off], "Y": [3.0 - off, 4.0 - off], "Gene_label": [2, 1]})
    res = sc.score_against_ground_truth(decoded, gt, ["g1", "g2"], voxel, shape, search_radius=0.05)
    # second pair: decoded z centre 0.15 is inside the support, GT z 0.01 is below z_min = 0.15 -> dropped
    assert (res["True Positives"], res["False Positives"], res["False Negatives"]) == (1, 1, 0)


def test_pixeldecoder_filter_driver_methods(golden):
    PD = pkg("PixelDecoder")
    g = golden("postdecode")
    dec = PD.PixelDecoder.__new__(PD.PixelDecoder)
    dec._df_barcodes_loaded = _table(g)
    dec._blank_count, dec._barcode_count, dec._verbose, dec._is_3D = 10, 140, 0, True
    dec._apply_filter_method("blank_fraction", 0.05, 0.05)
    np.testing.assert_array_equal(dec._df_filtered_barcodes.index.to_numpy(), g["auto_kept_rows"])
    assert dec._barcodes_filtered and dec._blank_fraction_filter_results["target_reached"]
    dec._df_filtered_barcodes["row_id"] = dec._df_filtered_barcodes.index.to_numpy()
    dec._remove_duplicates_within_tile(radius_xy=0.098, radius_z=0.31)
    np.testing.assert_array_equal(dec._df_filtered_barcodes["row_id"].to_numpy(), g["within_rows"])
    dec._remove_duplicates_in_tile_overlap()
    np.testing.assert_array_equal(dec._df_filtered_barcodes["row_id"].to_numpy(), g["overlap_rows"])
    with pytest.raises(ValueError, match="lr_fdr_target only applies"):
        PD.PixelDecoder._validate_filter_configuration("blank_fraction", 0.05, 0.1)
    with pytest.raises(ValueError, match="target_gross_misid_rate only applies"):
        PD.PixelDecoder._validate_filter_configuration("lr", 0.1, 0.05)
    with pytest.raises(ValueError, match="must be one of"):
        PD.PixelDecoder._validate_filter_configuration("other", 0.05, 0.05)


def _lr_table(g):
    base = np.load(str(__import__("conftest").GOLDEN / "postdecode.npz"))
    df = pd.DataFrame({"gene_id": base["gene_sorted"][base["gene_code"]]})
    for c in g.files:
        if not c.startswith(("3d_", "2d_", "noblank_", "is_blank")):
            df[c] = g[c]
    assert (df["gene_id"].str.startswith("Blank").to_numpy() == g["is_blank"]).all()
    return df


def test_lr_filter_matches_reference_run(golden):
    pd_mod = pkg("postdecode")
    PD = pkg("PixelDecoder")
    g = golden("postdecode_lr")
    table = _lr_table(g)
    for tag, is3d, target in (("3d", True, 0.05), ("2d", False, 0.2)):
        out, threshold = pd_mod.lr_filter(table, is_3d=is3d, blank_count=10, barcode_count=140, lr_fdr_target=target)
        np.testing.assert_array_equal(out.index.to_numpy(), g[f"{tag}_kept_rows"])
        np.testing.assert_allclose(out["predicted_probability"].to_numpy(), g[f"{tag}_probability"], rtol=1e-12)
        assert (out["cell_id"] == -1).all() and 0.0 <= threshold < 1.0
    few = table[~g["is_blank"]].iloc[:30]
    out, threshold = pd_mod.lr_filter(few, is_3d=True, blank_count=10, barcode_count=140)
    assert threshold is None
    np.testing.assert_array_equal(out.index.to_numpy(), g["noblank_kept_rows"])
    # driver method + FDR helper
    dec = PD.PixelDecoder.__new__(PD.PixelDecoder)
    dec._df_barcodes_loaded = table
    dec._blank_count, dec._barcode_count, dec._verbose, dec._is_3D = 10, 140, 0, True
    dec._apply_filter_method("lr", 0.05, 0.05)
    np.testing.assert_array_equal(dec._df_filtered_barcodes.index.to_numpy(), g["3d_kept_rows"])
    scored = dec._df_filtered_barcodes.copy()
    blank = scored["gene_id"].str.startswith("Blank")
    want = (blank.sum() / 10) / ((~blank).sum() / 130)
    assert PD.PixelDecoder._calculate_lr_fdr(scored, -1.0, 10, 140) == want
    assert PD.PixelDecoder._calculate_lr_fdr(scored[blank].copy(), 0.0, 10, 140) == np.inf
