"""Post-decode transcript filtering and duplicate removal.

Host-side table work, exactly as in the reference (pandas / NumPy / cKDTree on a few 10^5
rows; nothing here touches the GPU in the reference either):

* ``blank_fraction_filter``            PixelDecoder.py:3386-3846
* ``lr_filter`` / ``lr_fdr``             PixelDecoder.py:3848-4057 (scikit-learn, like the reference)
* ``remove_duplicates_in_tile_overlap`` PixelDecoder.py:4137-4177
* ``remove_duplicates_within_tile``     PixelDecoder.py:4179-4363

Semantics preserved: decile / integer-centred / linear bin edges with a ``nextafter`` upper
edge, float32 ``searchsorted(side="right") - 1`` binning (the reference bins on the device in
float32), the threshold sweep keeps the LAST threshold that meets the target (:3797-3801) and
otherwise the first minimum, overlap duplicates drop the larger ``(distance_min, row)``,
within-tile clusters keep the smallest ``(distance_min, row)``.
"""

from __future__ import annotations

from collections.abc import Sequence

import numpy as np
import pandas as pd
from scipy.spatial import cKDTree

_FEATURES = (("voxel_intensity", "magnitude_mean"), ("voxel_number", "area"), ("vector_distance", "distance_min"))


def _finish_edges(edges: np.ndarray, values: np.ndarray) -> np.ndarray:
    """Shared tail of the three automatic edge rules (:3519-3536, :3588-3609, :3626-3646)."""
    edges = np.unique(np.asarray(edges, dtype=float))
    edges = edges[np.isfinite(edges)]
    if edges.size < 2:
        centre = float(np.mean(values))
        edges = np.array([centre - 0.5, centre + 0.5], dtype=float)
    elif np.allclose(edges[0], edges[-1]):
        centre = float(edges[0])
        edges = np.array([centre - 0.5, centre + 0.5], dtype=float)
    edges[0] = min(edges[0], float(np.min(values)))
    edges[-1] = max(edges[-1], float(np.max(values)))
    edges[-1] = np.nextafter(edges[-1], np.inf)
    return edges


def _explicit_edges(bins: Sequence[float]) -> np.ndarray:
    edges = np.unique(np.asarray(bins, dtype=float))
    edges = edges[np.isfinite(edges)]
    if edges.size < 2:
        raise ValueError("Explicit histogram edges must contain at least two finite values.")
    edges[-1] = np.nextafter(edges[-1], np.inf)
    return edges


def _intensity_edges(values: np.ndarray) -> np.ndarray:
    """Deciles of the intensity (:3509-3536)."""
    return _finish_edges(np.quantile(values, np.linspace(0.0, 1.0, 11)), values)


def _voxel_number_edges(values: np.ndarray) -> np.ndarray:
    """Integer-centred bins: one per count when <= 10 distinct counts fit, else floored deciles (:3551-3609)."""
    lo, hi = int(np.floor(np.min(values))), int(np.ceil(np.max(values)))
    if hi - lo + 1 <= 10:
        edges = np.arange(lo - 0.5, hi + 1.5, 1.0)
    else:
        q = np.unique(np.floor(np.quantile(values, np.linspace(0.0, 1.0, 11))).astype(float))
        if q.size == 0:
            q = np.array([float(lo), float(hi + 1)])
        if q[0] > lo:
            q = np.insert(q, 0, float(lo))
        if q[-1] <= hi:
            q = np.append(q, float(hi + 1))
        edges = q - 0.5
    return _finish_edges(edges, values)


def _distance_edges(values: np.ndarray) -> np.ndarray:
    """Ten linear bins over the observed range (:3625-3646)."""
    return _finish_edges(np.linspace(float(np.min(values)), float(np.max(values)), 11), values)


def _empty_diagnostics(target: float) -> dict:
    return {
        "target_gross_misid_rate": float(target),
        "chosen_threshold": np.nan,
        "achieved_gross_misid_rate": np.inf,
        "target_reached": False,
        "all_histogram": np.zeros((0, 0, 0), dtype=np.int64),
        "blank_histogram": np.zeros((0, 0, 0), dtype=np.int64),
        "blank_fraction_histogram": np.zeros((0, 0, 0), dtype=float),
        "intensity_bins": np.array([], dtype=float),
        "voxel_number_bins": np.array([], dtype=float),
        "vector_distance_bins": np.array([], dtype=float),
        "threshold_sweep": pd.DataFrame(columns=["threshold", "gross_misid_rate", "kept_transcripts"]),
    }


def blank_fraction_filter(barcodes: pd.DataFrame, *, blank_count: int, barcode_count: int,
                          target_gross_misid_rate: float = 0.05, intensity_bins: Sequence[float] | None = None,
                          voxel_number_bins: Sequence[float] | None = None,
                          vector_distance_bins: Sequence[float] | None = None):
    """(filtered table with ``cell_id = -1``, diagnostics) - PixelDecoder.py:3386-3846.

    Transcripts are binned in (mean magnitude, area, minimum codeword distance); each bin's blank
    fraction is a threshold candidate; the kept set is every bin at or below the chosen threshold,
    where gross mis-identification = (blank kept / blank codewords) / (kept / all codewords).
    """
    missing = sorted({"gene_id", "magnitude_mean", "area", "distance_min"}.difference(barcodes.columns))
    if missing:
        raise ValueError("Blank-fraction filtering requires columns: " + ", ".join(missing)
                         + ". Re-decode transcripts with the exact caller.")
    annotated = barcodes.copy()
    for feature, column in _FEATURES:
        annotated[feature] = annotated[column].to_numpy(dtype=float, copy=False)
    annotated["is_blank"] = (annotated["gene_id"].astype("string").str.lower().str.startswith("blank", na=False)
                             .to_numpy(dtype=bool, copy=False))
    annotated["blank_fraction_bin"] = -1
    annotated["blank_fraction"] = np.nan
    annotated["blank_fraction_keep"] = False
    diagnostics = _empty_diagnostics(target_gross_misid_rate)

    def _done():
        filtered = annotated[annotated["blank_fraction_keep"]].copy()
        filtered["cell_id"] = -1
        return filtered, diagnostics

    if annotated.empty:
        diagnostics["reason"] = "no_transcripts"
        return _done()
    feats32 = [annotated[f].to_numpy(dtype=np.float32, copy=False) for f, _ in _FEATURES]
    is_blank = annotated["is_blank"].to_numpy(dtype=bool, copy=False)
    valid = np.all(np.isfinite(np.column_stack(feats32)), axis=1)
    if not valid.any():
        diagnostics["reason"] = "no_valid_features"
        return _done()
    if blank_count <= 0:
        annotated.loc[valid, "blank_fraction_keep"] = True
        diagnostics["reason"] = "no_blank_barcodes"
        return _done()
    if not is_blank[valid].any():
        annotated.loc[valid, "blank_fraction_keep"] = True
        diagnostics["reason"] = "no_blank_transcripts"
        return _done()

    values64 = [annotated.loc[valid, f].to_numpy(dtype=float, copy=False) for f, _ in _FEATURES]
    rules = (_intensity_edges, _voxel_number_edges, _distance_edges)
    explicit = (intensity_bins, voxel_number_bins, vector_distance_bins)
    edges = [_explicit_edges(e) if e is not None else rule(v) for e, rule, v in zip(explicit, rules, values64)]
    diagnostics["intensity_bins"], diagnostics["voxel_number_bins"], diagnostics["vector_distance_bins"] = edges

    # float32 edges and values, like the device-side binning of the reference (:3650-3680)
    bins = np.column_stack([np.searchsorted(e.astype(np.float32), v, side="right") - 1 for e, v in zip(edges, feats32)])
    in_range = valid.copy()
    for axis, (e, v) in enumerate(zip(edges, feats32)):
        in_range &= np.isfinite(v) & (bins[:, axis] >= 0) & (bins[:, axis] < len(e) - 1)
    if not in_range.any():
        diagnostics["reason"] = "no_transcripts_in_histogram_range"
        return _done()

    shape = tuple(len(e) - 1 for e in edges)
    all_hist = np.zeros(shape, dtype=np.int32)
    blank_hist = np.zeros(shape, dtype=np.int32)
    np.add.at(all_hist, tuple(bins[in_range].T), 1)
    blank_in_range = in_range & is_blank
    if blank_in_range.any():
        np.add.at(blank_hist, tuple(bins[blank_in_range].T), 1)
    fraction = np.full(shape, np.nan, dtype=np.float32)
    nonempty = all_hist > 0
    fraction[nonempty] = blank_hist[nonempty] / all_hist[nonempty]
    flat = np.ravel_multi_index(tuple(bins[in_range].T), dims=shape)
    flat_bins = np.full(len(annotated), -1, dtype=np.int64)
    flat_bins[in_range] = flat
    annotated["blank_fraction_bin"] = flat_bins
    annotated.loc[in_range, "blank_fraction"] = fraction.ravel()[flat]

    per_row = annotated["blank_fraction"].to_numpy(dtype=float, copy=False)
    sweep, chosen_threshold, achieved, chosen_keep, reached = [], np.nan, np.inf, np.zeros(len(annotated), bool), False
    for threshold in np.unique(fraction[nonempty]):
        keep = in_range & (per_row <= float(threshold))
        if blank_count <= 0 or barcode_count <= 0 or not keep.any():
            gross = np.inf
        else:
            gross = (np.count_nonzero(keep & is_blank) / float(blank_count)) / (
                np.count_nonzero(keep) / float(barcode_count))
        sweep.append({"threshold": float(threshold), "gross_misid_rate": float(gross),
                      "kept_transcripts": int(np.count_nonzero(keep))})
        if gross <= target_gross_misid_rate:  # the last (loosest) threshold that meets the target wins
            chosen_threshold, achieved, chosen_keep, reached = float(threshold), float(gross), keep.copy(), True
    if not sweep:
        diagnostics["reason"] = "no_nonempty_histogram_bins"
        return _done()
    sweep_df = pd.DataFrame(sweep)
    if not reached:
        best = int(sweep_df["gross_misid_rate"].argmin())
        chosen_threshold = float(sweep_df.loc[best, "threshold"])
        achieved = float(sweep_df.loc[best, "gross_misid_rate"])
        chosen_keep = in_range & (per_row <= chosen_threshold)
    annotated["blank_fraction_keep"] = chosen_keep
    diagnostics.update({"chosen_threshold": chosen_threshold, "achieved_gross_misid_rate": achieved,
                        "target_reached": reached, "all_histogram": all_hist, "blank_histogram": blank_hist,
                        "blank_fraction_histogram": fraction, "threshold_sweep": sweep_df})
    return _done()


def lr_fdr(df: pd.DataFrame, threshold: float, blank_count: int, barcode_count: int) -> float:
    """(blank calls / blank codewords) / (coding calls / coding codewords) among rows whose predicted
    probability exceeds ``threshold`` (all rows for a negative threshold) - PixelDecoder.py:3848-3905."""
    blank = df["gene_id"].astype("string").str.lower().str.startswith("blank", na=False)
    if threshold >= 0:
        positive = df["predicted_probability"] > threshold
        df["prediction"] = positive
        coding, noncoding = int((~blank & positive).sum()), int((blank & positive).sum())
    else:
        coding, noncoding = int((~blank).sum()), int(blank.sum())
    if coding <= 0:
        return np.inf
    return (noncoding / blank_count) / (coding / (barcode_count - blank_count))


_LR_FEATURES_2D = ["area", "signal_mean", "s-b_mean", "distance_min", "magnitude_mean", "inertia_tensor_eigvals-0",
                   "inertia_tensor_eigvals-1"]


def lr_filter(barcodes: pd.DataFrame, *, is_3d: bool, blank_count: int, barcode_count: int,
              lr_fdr_target: float = 0.05):
    """(filtered table with ``cell_id = -1``, chosen probability threshold or None) - PixelDecoder.py:3907-4057.

    A logistic regression (liblinear, seed 42) is trained to tell coding from blank calls on a
    class-balanced 90 % split of standardised features; the probability threshold is the first of
    0, 0.1, ... meeting the FDR target, refined on a 0.01 grid around it.  With fewer than two blank
    calls nothing is filtered."""
    from sklearn.linear_model import LogisticRegression
    from sklearn.model_selection import train_test_split
    from sklearn.preprocessing import StandardScaler

    df = barcodes.copy()
    df["X"] = ~df["gene_id"].astype("string").str.lower().str.startswith("blank", na=False).to_numpy(dtype=bool)
    features = _LR_FEATURES_2D + (["inertia_tensor_eigvals-2"] if is_3d else [])
    columns = ["X"] + features
    coding, blanks = df[df["X"]][columns], df[~df["X"]][columns]
    if len(blanks) <= 1:
        out = barcodes.copy()
        out["cell_id"] = -1
        return out, None
    balanced = pd.concat([coding.sample(n=len(blanks), random_state=42), blanks])
    x_train, _x_test, y_train, _y_test = train_test_split(balanced.drop("X", axis=1), balanced["X"], test_size=0.1,
                                                          random_state=42)
    scaler = StandardScaler()
    model = LogisticRegression(solver="liblinear", random_state=42)
    model.fit(scaler.fit_transform(x_train), y_train)
    df["predicted_probability"] = model.predict_proba(scaler.transform(df[features]))[:, 1]

    def first_meeting(grid, default):
        for threshold in grid:
            if lr_fdr(df, threshold, blank_count, barcode_count) <= lr_fdr_target:
                return threshold
        return default

    coarse = first_meeting(np.arange(0, 1, 0.1), 0)
    fine = first_meeting(np.arange(coarse - 0.1, coarse + 0.1, 0.01), coarse)
    out = df[df["predicted_probability"] > fine].copy()
    out["cell_id"] = -1
    return out, float(fine)


def remove_duplicates_in_tile_overlap(barcodes: pd.DataFrame, radius: float = 0.75):
    """(table, dropped distance_min values): for every pair of transcripts from DIFFERENT tiles within
    ``radius`` um, drop the one with the larger ``(distance_min, row)`` (PixelDecoder.py:4137-4177)."""
    df = barcodes.reset_index(drop=True)
    coords = df[["global_z", "global_y", "global_x"]].values
    tiles = df["tile_idx"].values
    dmin = df["distance_min"].to_numpy(dtype=float, copy=False)
    drop, dropped = set(), []
    for i, j in cKDTree(coords).query_pairs(radius):
        if tiles[i] == tiles[j]:
            continue
        loser = j if (dmin[i], i) <= (dmin[j], j) else i
        drop.add(loser)
        dropped.append(dmin[loser])
    df = df.drop(drop).reset_index(drop=True)
    return df, np.asarray(dropped, dtype=float)


def _components(n: int, pairs) -> dict:
    """Connected components of an undirected pair list (union-find with path halving)."""
    parent = np.arange(n)
    rank = np.zeros(n, dtype=np.int8)

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for a, b in pairs:
        ra, rb = find(a), find(b)
        if ra == rb:
            continue
        if rank[ra] < rank[rb]:
            parent[ra] = rb
        else:
            parent[rb] = ra
            if rank[ra] == rank[rb]:
                rank[ra] += 1
    comps: dict[int, list[int]] = {}
    for i in range(n):
        comps.setdefault(find(i), []).append(i)
    return comps


def remove_duplicates_within_tile(barcodes: pd.DataFrame, radius_xy: float = 0.1, radius_z: float = 0.50):
    """(table, dropped row indices): collapse same-gene detections of one tile that sit within
    ``radius_xy`` laterally on different planes at most ``radius_z`` apart; each cluster keeps its
    smallest ``(distance_min, row)`` (PixelDecoder.py:4179-4363)."""
    if barcodes is None or barcodes.empty or len(barcodes) < 2:
        return barcodes, []
    df = barcodes.reset_index(drop=True)
    coords = df[["global_z", "global_y", "global_x"]].to_numpy(dtype=float, copy=False)
    tiles = df["tile_idx"].to_numpy()
    genes = df["gene_id"].to_numpy()
    dmin = df["distance_min"].to_numpy(dtype=float, copy=False)
    drop: set[int] = set()
    for tile in np.unique(tiles):
        rows = np.flatnonzero(tiles == tile)
        if rows.size < 2:
            continue
        z, g = coords[rows, 0], genes[rows]
        pairs = [(i, j) for i, j in cKDTree(coords[rows, 1:3]).query_pairs(r=radius_xy)
                 if 0.0 < abs(z[i] - z[j]) <= radius_z and g[i] == g[j]]
        if not pairs:
            continue
        for members in _components(rows.size, pairs).values():
            if len(members) < 2:
                continue
            glob = rows[np.asarray(members)]
            best = glob[np.lexsort((glob, dmin[glob]))][0]
            drop.update(int(r) for r in glob if r != best)
    if drop:
        df = df.drop(index=list(drop)).reset_index(drop=True)
    return df, sorted(drop)
