"""Restatement of PixelDecoder._extract_barcodes (PixelDecoder.py:2908-3201).

ORACLE / TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.

The reference delegates the arithmetic to cuCIM / scikit-image (``label``,
``remove_small_objects``, ``regionprops_table``).  No reference test exercises
``_extract_barcodes``; the library arithmetic restated here is pinned bit for bit by
``tests/golden/skimage_regionprops.npz`` (real scikit-image 0.18.3,
``tests/golden/make_golden_skimage.py``), the table post-processing is a restatement.
What it fixes:

* ``label(background=-1, connectivity=3)``: 26-connected regions of *equal* codeword id,
  numbered in raster order of each region's first voxel; 2-D mode labels each plane with
  8-connectivity and a running offset (:2946-2972);
* regions with more than ``maximum_pixels`` voxels, then fewer than ``minimum_pixels``,
  are dropped (:2976-2989);
* ``regionprops_table``: ``area``, ``centroid`` (float64 voxel means), ``intensity_mean``
  per channel (``np.mean`` of the region's values: float16 in -> float32 sum -> float16
  out, exactly NumPy's rule), ``inertia_tensor_eigvals`` (central second moments / area,
  ``eigvalsh``, clipped at 0, descending), ``intensity_min`` of the float16 distance image;
* the table post-processing of :3062-3177 verbatim.
"""

from __future__ import annotations

import numpy as np
import pandas as pd

from . import build as _build
from .decode_ref import warp_pixel


def label_equal(decoded: np.ndarray, is_3d: bool = True) -> tuple[np.ndarray, int]:
    """int32 labels (0 = background) of equal-valued regions, scikit-image numbering."""
    img = np.ascontiguousarray(decoded, dtype=np.int16)
    lib = _build.lib()
    if is_3d:
        labels = np.zeros(img.shape, dtype=np.int32)
        n = lib.m3d_oracle_label_equal(img.ctypes.data, *img.shape, -1, 3, labels.ctypes.data)
        return labels, int(n)
    labels = np.zeros(img.shape, dtype=np.int32)
    offset = 0
    for z in range(img.shape[0]):
        plane = np.ascontiguousarray(img[z])
        lab = np.zeros(plane.shape, dtype=np.int32)
        n = lib.m3d_oracle_label_equal(plane.ctypes.data, 1, *plane.shape, -1, 2, lab.ctypes.data)
        if n:
            labels[z] = np.where(lab != 0, lab + offset, 0)
            offset += int(n)
    return labels, offset


def inertia_eigvals(coords_zyx: np.ndarray) -> np.ndarray:
    """scikit-image ``inertia_tensor_eigvals`` of a binary region given its voxel coordinates."""
    c = coords_zyx.astype(np.float64)
    mu0 = float(c.shape[0])
    d = c - c.mean(axis=0)
    mu2 = (d * d).sum(axis=0)
    T = np.zeros((3, 3), dtype=np.float64)
    for i in range(3):
        T[i, i] = (mu2.sum() - mu2[i]) / mu0
    for i, j in ((0, 1), (0, 2), (1, 2)):
        T[i, j] = T[j, i] = -(d[:, i] * d[:, j]).sum() / mu0
    ev = np.clip(np.linalg.eigvalsh(T), 0, None)
    return np.sort(ev)[::-1]


def region_table(labels: np.ndarray, intensity_zyxb: np.ndarray, magnitude: np.ndarray,
                 distance: np.ndarray) -> pd.DataFrame:
    """The merged regionprops tables of :2991-3060 for every non-empty label."""
    flat = labels.ravel()
    fg = np.nonzero(flat)[0]
    order = np.argsort(flat[fg], kind="stable")
    fg = fg[order]
    lab_sorted = flat[fg]
    uniq, start = np.unique(lab_sorted, return_index=True)
    stop = np.append(start[1:], lab_sorted.size)
    zyx = np.stack(np.unravel_index(fg, labels.shape), axis=1)
    inten = intensity_zyxb.reshape(-1, intensity_zyxb.shape[-1])
    mag = magnitude.ravel()
    dist = distance.ravel().astype(np.float32)
    B = inten.shape[1]
    rows = []
    for lab, a, b in zip(uniq, start, stop):
        idx = fg[a:b]
        co = zyx[a:b]
        cen = co.astype(np.float64).mean(axis=0)
        row = {"label": int(lab), "area": float(b - a),
               "centroid-0": cen[0], "centroid-1": cen[1], "centroid-2": cen[2]}
        means = np.mean(inten[idx], axis=0)  # float16 -> float16 via float32 (NumPy rule)
        for k in range(B):
            row[f"intensity_mean-{k}"] = means[k]
        ev = inertia_eigvals(co)
        for k in range(3):
            row[f"inertia_tensor_eigvals-{k}"] = ev[k]
        row["distance_min"] = dist[idx].min()
        row["magnitude_mean"] = np.mean(mag[idx])
        rows.append(row)
    return pd.DataFrame(rows)


def extract_barcodes(
    decoded: np.ndarray,
    intensity_bzyx: np.ndarray,
    magnitude: np.ndarray,
    distance: np.ndarray,
    codebook_matrix: np.ndarray,
    gene_ids,
    transcript_distance_threshold: float,
    *,
    is_3d: bool = True,
    minimum_pixels: float = 3,
    maximum_pixels: int = 500,
    tile_idx: int = 0,
    z_offset: float | None = None,
    spacing=(1.0, 1.0, 1.0),
    origin=(0.0, 0.0, 0.0),
    affine=np.eye(4),
    camera_to_stage=np.eye(4),
) -> pd.DataFrame:
    """Whole ``_extract_barcodes`` on host arrays (PixelDecoder.py:2908-3201)."""
    n_bits = intensity_bzyx.shape[0]
    labels, _ = label_equal(decoded, is_3d)
    counts = np.bincount(labels.ravel())
    large = np.nonzero(counts > maximum_pixels)[0]
    large = large[large != 0]
    if large.size:
        labels[np.isin(labels, large)] = 0
    max_size = max(int(minimum_pixels) - 1, 0)  # remove_small_objects(max_size=...) (:2987-2989)
    counts = np.bincount(labels.ravel())
    small = np.nonzero(counts <= max_size)[0]
    small = small[small != 0]
    if small.size:
        labels[np.isin(labels, small)] = 0

    df = region_table(labels, np.moveaxis(intensity_bzyx, 0, -1), magnitude, distance)
    if df.empty:
        return df
    df = df[df["area"] > 0.1].reset_index(drop=True)
    flat_l, flat_d = labels.ravel(), decoded.ravel()
    first = {}
    order = np.argsort(flat_l, kind="mergesort")
    uniq, first_idx = np.unique(flat_l[order], return_index=True)
    label_to_id = np.full(int(flat_l.max()) + 1, -1, dtype=np.int16)
    label_to_id[uniq] = flat_d[order][first_idx]
    label_to_id[0] = -1
    df["decoded_id"] = label_to_id[df["label"].to_numpy(dtype=np.int64)].astype(np.int32)
    df = df[df["decoded_id"] >= 0].reset_index(drop=True)
    df["barcode_id"] = df["decoded_id"].astype(np.int32) + 1
    df["gene_id"] = [gene_ids[x] for x in df["decoded_id"].to_numpy(dtype=np.int32)]
    df["tile_idx"] = tile_idx
    cb = np.asarray(codebook_matrix).astype(bool)
    on_bits = np.argsort(~cb, axis=1, kind="stable")[:, :4].astype(np.int32) + 1  # set semantics
    on_sel = on_bits[df["decoded_id"].to_numpy(dtype=np.int32)]
    for k in range(4):
        df[f"on_bit_{k + 1}"] = on_sel[:, k]
    df = df.rename(columns={"centroid-0": "z", "centroid-1": "y", "centroid-2": "x"})
    if z_offset is not None:
        df["z"] = float(z_offset) + df["z"]
    df["tile_z"] = np.round(df["z"], 0).astype(int)
    df["tile_y"] = np.round(df["y"], 0).astype(int)
    df["tile_x"] = np.round(df["x"], 0).astype(int)
    pts = df[["z", "y", "x"]].to_numpy()
    for i in range(pts.shape[0]):
        pts[i, :] = warp_pixel(pts[i, :].copy(), np.asarray(spacing, np.float32),
                               np.asarray(origin, np.float32), np.asarray(affine, np.float32),
                               np.asarray(camera_to_stage, np.float32))
    df["global_z"] = np.round(pts[:, 0], 2)
    df["global_y"] = np.round(pts[:, 1], 2)
    df["global_x"] = np.round(pts[:, 2], 2)
    for i in range(1, n_bits + 1):
        df = df.rename(columns={f"intensity_mean-{i - 1}": f"bit{i:02d}_mean_intensity"})
    bit_cols = [f"bit{i:02d}_mean_intensity" for i in range(1, n_bits + 1)]
    bit_means = df[bit_cols].to_numpy(dtype=np.float64)
    total = bit_means.sum(axis=1)
    on0 = df[["on_bit_1", "on_bit_2", "on_bit_3", "on_bit_4"]].to_numpy(dtype=np.int32) - 1
    signal = np.take_along_axis(bit_means, on0, axis=1).sum(axis=1)
    df["signal_mean"] = signal / 4.0
    df["bkd_mean"] = (total - signal) / float(n_bits - 4)
    df["s-b_mean"] = df["signal_mean"] - df["bkd_mean"]
    df = df.drop(columns=["label", "decoded_id"])
    df = df[df["distance_min"] <= transcript_distance_threshold].reset_index(drop=True)
    return df
