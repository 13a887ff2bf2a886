"""F1 scorer of the simulation test matrix.

``calculate_F1_with_radius`` mirrors cli/statphysbio_simulation/calculate_F1.py:19-200: greedy
closest-first one-to-one matching of decoded and ground-truth transcripts of the same gene within
``radius``.  ``score_against_ground_truth`` is the coordinate bookkeeping of the CLI command
(:239-268) without its datastore plumbing: decoded z moved to the voxel centre, the simulator's
X / Y swap, re-centring of the ground truth on the image, and the z-support crop.
"""

from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree


def _metrics(tp: int, fp: int, fn: int) -> dict:
    precision = tp / (tp + fp) if (tp + fp) else 0.0
    recall = tp / (tp + fn) if (tp + fn) else 0.0
    f1 = 2 * precision * recall / (precision + recall) if (precision + recall) else 0.0
    return {"F1 Score": f1, "Precision": precision, "Recall": recall, "True Positives": tp,
            "False Positives": fp, "False Negatives": fn}


def calculate_F1_with_radius(qi2lab_coords, qi2lab_gene_ids, gt_coords, gt_gene_ids, radius: float):
    """(metrics, true-positive coords, false-positive coords, false-negative coords)."""
    q_xyz = np.asarray(qi2lab_coords, float)
    g_xyz = np.asarray(gt_coords, float)
    q_gene = np.asarray(qi2lab_gene_ids).astype(str)
    g_gene = np.asarray(gt_gene_ids).astype(str)
    nq, ng = q_xyz.shape[0], g_xyz.shape[0]
    none = np.empty((0, 3), float)
    if nq == 0 and ng == 0:
        return ({"F1 Score": 1.0, "Precision": 1.0, "Recall": 1.0, "True Positives": 0, "False Positives": 0,
                 "False Negatives": 0}, none, none.copy(), none.copy())
    if nq == 0:
        return ({"F1 Score": 0.0, "Precision": 0.0, "Recall": 0.0, "True Positives": 0, "False Positives": 0,
                 "False Negatives": int(ng)}, none, none.copy(), g_xyz.copy())
    if ng == 0:
        return ({"F1 Score": 0.0, "Precision": 0.0, "Recall": 0.0, "True Positives": 0,
                 "False Positives": int(nq), "False Negatives": 0}, none, q_xyz.copy(), none.copy())

    # candidate pairs within the radius, gene by gene (sorted gene order, like np.intersect1d)
    pq, pg, pd_ = [], [], []
    for gene in np.intersect1d(np.unique(q_gene), np.unique(g_gene)):
        qi, gi = np.flatnonzero(q_gene == gene), np.flatnonzero(g_gene == gene)
        if qi.size == 0 or gi.size == 0:
            continue
        coo = cKDTree(q_xyz[qi]).sparse_distance_matrix(cKDTree(g_xyz[gi]), max_distance=radius,
                                                        output_type="coo_matrix")
        if coo.nnz == 0:
            continue
        pq.append(qi[coo.row])
        pg.append(gi[coo.col])
        pd_.append(coo.data)
    q_used, g_used = np.zeros(nq, dtype=bool), np.zeros(ng, dtype=bool)
    matched = []
    if pq:
        pq, pg = np.concatenate(pq), np.concatenate(pg)
        order = np.argsort(np.concatenate(pd_), kind="stable")  # closest first, stable on ties
        for a, b in zip(pq[order], pg[order]):
            if q_used[a] or g_used[b] or q_gene[a] != g_gene[b]:
                continue
            q_used[a] = g_used[b] = True
            matched.append(a)
    matched = np.asarray(matched, dtype=int)
    tp = int(matched.size)
    tp_coords = q_xyz[matched] if tp else none
    return (_metrics(tp, int((~q_used).sum()), int((~g_used).sum())), tp_coords, q_xyz[~q_used], g_xyz[~g_used])


def score_against_ground_truth(decoded_spots, gt_spots, gene_ids, voxel_size_zyx_um, image_shape_zyx,
                               search_radius: float = 1.0) -> dict:
    """Metrics of a filtered decoded table against the simulator's ``GT_spots.csv`` table
    (columns Z, X, Y, Gene_label) - calculate_F1.py:239-268."""
    vz, vy, vx = (float(v) for v in voxel_size_zyx_um)
    q = decoded_spots[["global_z", "global_y", "global_x"]].to_numpy(dtype=float).copy()
    q[:, 0] += vz / 2                                                 # voxel index -> voxel centre
    q_gene = decoded_spots["gene_id"].to_numpy()
    g = gt_spots[["Z", "X", "Y"]].to_numpy(dtype=float)                # the simulator swaps X and Y
    g = g + [0, (1.0 * image_shape_zyx[-2] / 2) * vy - vy / 2, (1.0 * image_shape_zyx[-1] / 2) * vx - vx / 2]
    g_gene = np.array(gene_ids)[gt_spots["Gene_label"].to_numpy(dtype=int) - 1]
    z_lo, z_hi = vz / 2, (image_shape_zyx[-3] - 0.5) * vz
    qk = (q[:, 0] >= z_lo) & (q[:, 0] <= z_hi)
    gk = (g[:, 0] >= z_lo) & (g[:, 0] <= z_hi)
    return calculate_F1_with_radius(q[qk], q_gene[qk], g[gk], g_gene[gk], search_radius)[0]
