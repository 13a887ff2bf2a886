"""Synthetic fixed/moving pair with a known smooth deformation, for the flow-recovery tests.

Same experiment as the reference's tolerance test for the estimator
(tests/test_sofima_deformable_registration.py:221-527): point emitters and blobs on a 3x finer
grid, moved by a known coarse flow, blurred and binned to the camera grid.  Written from the
test's description (sizes, counts, amplitudes, thresholds), not from its text.
"""

from __future__ import annotations

import numpy as np
from scipy.ndimage import gaussian_filter, map_coordinates

UPSAMPLE = 3
CAMERA_SHAPE = (32, 128, 128)
PATCH = (10, 32, 32)
STRIDE = (5.0, 16.0, 16.0)
BOX_START_XYZ = (PATCH[2] / 2.0, PATCH[1] / 2.0, PATCH[0] / 2.0)


def _deposit(volume, pts, weights):
    """Trilinear deposit of weighted points."""
    lo = np.floor(pts).astype(np.int64)
    fr = pts - lo
    for corner in np.ndindex(2, 2, 2):
        idx = lo + np.asarray(corner)
        w = weights.copy()
        for ax in range(3):
            w = w * (fr[:, ax] if corner[ax] else 1.0 - fr[:, ax])
        ok = np.all((idx >= 0) & (idx < np.asarray(volume.shape)), axis=1)
        np.add.at(volume, tuple(idx[ok].T), w[ok])


def known_flow(amplitude=10.0):
    grid = tuple(int((n - p) // s) + 1 for n, p, s in zip(CAMERA_SHAPE, PATCH, STRIDE))
    g = np.indices(grid, dtype=np.float32)
    c = [(n - 1) / 2 for n in grid]
    zn, yn, xn = [(g[i] - c[i]) / max(c[i], 1) for i in range(3)]
    f = np.zeros((3,) + grid, dtype=np.float32)
    f[0] = amplitude * (0.8 * xn + 0.2 * yn)
    f[1] = amplitude * (-0.7 * yn + 0.15 * xn)
    f[2] = amplitude * (0.4 * zn + 0.05 * xn)
    return f


def sample_flow(flow_xyz, pts_zyx, stride=STRIDE, box_xyz=BOX_START_XYZ):
    """displacement (z, y, x) of camera-grid points under a flow field (order 1, nearest)."""
    c = np.stack([(pts_zyx[:, 0] - box_xyz[2]) / stride[0], (pts_zyx[:, 1] - box_xyz[1]) / stride[1],
                  (pts_zyx[:, 2] - box_xyz[0]) / stride[2]])
    d = np.stack([map_coordinates(flow_xyz[ch], c, order=1, mode="nearest") for ch in range(3)], axis=1)
    return d[:, ::-1].astype(np.float32)


def make_pair(amplitude=10.0, seed=0):
    """(fixed, moving, known_flow_xyz)."""
    rng = np.random.default_rng(seed)
    hi = np.asarray([n * UPSAMPLE for n in CAMERA_SHAPE], dtype=np.float32)
    margin = np.asarray((8, 18, 18), dtype=np.float32)
    pts = rng.uniform(margin, hi - margin, size=(3500, 3)).astype(np.float32)
    wts = rng.uniform(10, 100, 3500).astype(np.float32)
    blobs = []
    for _ in range(80):
        blobs.append((rng.uniform(margin, hi - margin).astype(np.float32), np.float32(rng.uniform(30, 100)),
                      np.asarray([rng.uniform(3, 7), rng.uniform(4, 10), rng.uniform(4, 10)], dtype=np.float32)))
    flow = known_flow(amplitude)

    def render(points, centers):
        r = np.random.default_rng(seed + 1)
        vol = np.zeros(tuple(int(v) for v in hi), dtype=np.float32)
        _deposit(vol, points, wts)
        for (c0, amp, sig), c in zip(blobs, centers):
            s = c + r.normal(0.0, sig, size=(96, 3)).astype(np.float32)
            _deposit(vol, s, np.full(96, amp / 96.0, dtype=np.float32))
        vol = gaussian_filter(vol + 1.0, sigma=(3.6, 3.3, 3.3))
        zs = vol[UPSAMPLE // 2::UPSAMPLE]
        z, y, x = zs.shape
        return zs.reshape(z, y // UPSAMPLE, UPSAMPLE, x // UPSAMPLE, UPSAMPLE).mean(axis=(2, 4)).astype(np.float32)

    centers = np.stack([b[0] for b in blobs])
    fixed = render(pts, centers)
    moved_pts = pts + sample_flow(flow, pts / UPSAMPLE) * UPSAMPLE
    moved_centers = centers + sample_flow(flow, centers / UPSAMPLE) * UPSAMPLE
    moving = render(moved_pts.astype(np.float32), moved_centers.astype(np.float32))
    return fixed, moving, flow


def expected_on_grid(estimated_shape, metadata, flow):
    """The known flow resampled on the estimator's own grid."""
    st = [float(v) for v in metadata["map_stride_zyx_px"]]
    bx = [float(v) for v in metadata["map_box_start_xyz_px"]]
    g = np.indices(estimated_shape, dtype=np.float32)
    c = np.stack([(bx[2] + g[0] * st[0] - BOX_START_XYZ[2]) / STRIDE[0],
                  (bx[1] + g[1] * st[1] - BOX_START_XYZ[1]) / STRIDE[1],
                  (bx[0] + g[2] * st[2] - BOX_START_XYZ[0]) / STRIDE[2]])
    return np.stack([map_coordinates(flow[ch], c, order=1, mode="nearest") for ch in range(3)])


def check_recovery(estimated, metadata, flow):
    """The reference test's acceptance thresholds (test_sofima_deformable_registration.py:515-527)."""
    exp = expected_on_grid(estimated.shape[1:], metadata, flow)
    err = estimated - exp
    rmse = np.sqrt(np.mean(err ** 2, axis=(1, 2, 3)))
    assert metadata["status"] == "ok", metadata["status"]
    assert metadata["valid_flow_vectors"] > 400
    assert metadata["residual_iterations"] == 2
    assert estimated.shape[0] == 3
    assert np.sqrt(np.mean(err ** 2)) < 0.75
    assert np.sqrt(np.mean(err[:, 1:-1, 1:-1, 1:-1] ** 2)) < 0.75
    assert np.all(rmse < np.asarray([0.7, 0.7, 0.75])), rmse
    assert estimated[0].min() <= -9.5 and estimated[0].max() >= 9.5
    assert estimated[1].min() <= -7.5 and estimated[1].max() >= 7.5
    assert estimated[2].min() <= -4.5 and estimated[2].max() >= 3.5
    return rmse
