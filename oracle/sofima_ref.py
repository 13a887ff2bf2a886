"""Restatement of the residual flow estimator (utils/sofima_registration.py:9-794).

ORACLE / TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.

Two kinds of code live here:

* the reference's own glue - patch/step resolution (:48-77), axial stabilisation (:80-146),
  normalised MSE (:149-175), flow composition (:178-240), sub-pixel refinement (:243-417),
  median initialiser (:420-452), the residual-pass driver (:524-757).  Pinned by
  ``tests/golden/sofima_glue.npz`` (reference functions run with scipy standing in for
  ``jax.scipy.ndimage.map_coordinates``).
* the arithmetic the reference delegates to the third-party package ``sofima`` (Google
  Research, Apache-2.0; unpinned ``sofima`` dependency in the reference's pyproject, NOT in
  the reference tree): ``flow_field.JAXMaskedXCorrWithStatsCalculator.flow_field``,
  ``flow_utils.clean_flow``, ``mesh.relax_mesh`` + ``mesh.elastic_mesh_3d``.  Restated from
  the published algorithm:
    - per patch pair, mean-subtracted full linear cross-correlation (mask-free path of
      ``masked_xcorr``), integer peak = flow vector (post -> pre), peak sharpness = peak / min
      within ``peak_radius`` of it, peak ratio = peak / best other local maximum (non-maximum
      suppression +-``peak_min_distance``, relative threshold 0.5) outside that radius,
      0 when there is none;
    - ``clean_flow``: drop |sharpness| < min, 0 < |ratio| < min, any |component| > max
      magnitude, any component further than ``max_deviation`` from its 3x3x3 median;
    - elastic mesh: Hookean springs along the 3 axes and 6 face diagonals of the node grid
      (rest length from the stride, spring constant k * stride_x / rest length), zero-length
      springs k0 to every finite measured vector, force components capped, integrated with
      velocity-Verlet + FIRE (Bitzek 2006; f_inc 1.1, f_dec 0.5, alpha 0.1, f_alpha 0.99,
      n_min 5) in blocks of ``num_iters``; a block that ends with max|v| and max|F| < ``stop_v_max`` raises
      the force cap tenfold, and stops the run once the cap has reached ``final_cap``.  (The ramp
      factor is this project's choice - the library's is not known here; the answer is the
      mesh equilibrium, which does not depend on it.)
  **parity unpinned** for this part: the only anchor is the reference's tolerance test
  (tests/test_sofima_deformable_registration.py:221-527: status ok, > 400 valid vectors,
  two residual passes, flow RMSE < 0.75 px), reproduced in tests/test_oracle_sofima.py.
"""

from __future__ import annotations

import itertools
from dataclasses import asdict, dataclass
from typing import Any

import numpy as np
import scipy.fft as sfft
import scipy.ndimage as sndi



@dataclass(frozen=True)
class SofimaRegistrationConfig:
    """sofima_registration.py:9-45 (same fields and defaults)."""

    residual_iterations: int = 2
    patch_size_zyx: tuple = (10, 32, 32)
    minimum_patch_size_px: int = 4
    step_divisor: int = 2
    peak_min_distance: int = 2
    peak_radius: int = 8
    batch_size: int = 32
    max_masked: float = 0.75
    min_peak_ratio: float = 1.2
    min_peak_sharpness: float = 1.2
    max_magnitude: float = 30.0
    max_deviation: float = 5.0
    max_local_z_displacement_px: float = 5.0
    subpixel_offsets: tuple = (-0.5, 0.0, 0.5)
    subpixel_batch_size: int = 32
    normalization_epsilon: float = 1e-6
    mesh_dt: float = 0.001
    mesh_gamma: float = 0.0
    mesh_k0: float = 1.0
    mesh_k: float = 0.01
    mesh_num_iters: int = 1000
    mesh_max_iters: int = 20000
    mesh_stop_v_max: float = 0.001
    mesh_dt_max: float = 100.0
    mesh_start_cap: float = 0.1
    mesh_final_cap: float = 10.0

    def as_metadata(self) -> dict:
        m = asdict(self)
        m["patch_size_zyx"] = [int(v) for v in self.patch_size_zyx]
        m["subpixel_offsets"] = [float(v) for v in self.subpixel_offsets]
        return m


# FIRE constants and the cap ramp of the published mesh integrator
FIRE_F_INC, FIRE_F_DEC, FIRE_ALPHA, FIRE_F_ALPHA, FIRE_N_MIN = 1.1, 0.5, 0.1, 0.99, 5
CAP_UPSCALE = 10.0
PEAK_THRESHOLD_REL = 0.5
# spring directions (x, y, z) of the 3-D mesh: axes and face diagonals
MESH_LINKS_XYZ = ((1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, -1, 0), (1, 0, 1), (1, 0, -1),
                  (0, 1, 1), (0, 1, -1))


def resolve_patch_and_step(shape_zyx, config):
    """sofima_registration.py:48-77."""
    patch = tuple(int(max(config.minimum_patch_size_px, min(n, p)))
                  for n, p in zip(shape_zyx, config.patch_size_zyx))
    step = tuple(max(1, p // int(config.step_divisor)) for p in patch)
    return patch, step


def flow_grid_shape(shape_zyx, patch_zyx, step_zyx):
    return tuple((int(n) - int(p)) // int(s) + 1 for n, p, s in zip(shape_zyx, patch_zyx, step_zyx))


def xcorr_fft_shape(patch_zyx):
    return tuple(int(sfft.next_fast_len(2 * int(p) - 1, real=False)) for p in patch_zyx)


def patch_xcorr_full(pre_patch, post_patch):
    """Full linear cross-correlation c[d] = sum_u post(u) pre(u + d), d in [-(p-1), p-1], of the
    mean-subtracted patches; index d + (p-1)."""
    a = pre_patch.astype(np.float32) - np.float32(pre_patch.astype(np.float32).mean())
    b = post_patch.astype(np.float32) - np.float32(post_patch.astype(np.float32).mean())
    p = a.shape
    F = xcorr_fft_shape(p)
    xc = sfft.ifftn(sfft.fftn(a, F) * np.conj(sfft.fftn(b, F))).real.astype(np.float32)
    idx = [np.arange(-(n - 1), n) % f for n, f in zip(p, F)]
    return xc[np.ix_(*idx)]


def peak_stats(full, min_distance, peak_radius, threshold_rel=PEAK_THRESHOLD_REL):
    """(peak index zyx, sharpness, ratio) of one correlation volume."""
    pk = np.unravel_index(int(np.argmax(full)), full.shape)
    vmax = np.float32(full[pk])
    win = tuple(slice(max(0, c - peak_radius), min(n, c + peak_radius + 1)) for c, n in zip(pk, full.shape))
    vmin = np.float32(full[win].min())
    with np.errstate(divide="ignore", invalid="ignore"):
        sharp = np.float32(vmax) / np.float32(vmin)
    mx = sndi.maximum_filter(full, size=2 * min_distance + 1, mode="constant", cval=-np.inf)
    cand = (mx == full) & (full > np.float32(threshold_rel) * vmax)
    cand[win] = False
    if cand.any():
        with np.errstate(divide="ignore", invalid="ignore"):
            ratio = np.float32(vmax) / np.float32(full[cand].max())
    else:
        ratio = np.float32(0.0)
    return pk, sharp, ratio


def patch_flow_field(pre_image, post_image, patch_zyx, step_zyx, min_distance=2, peak_radius=8):
    """[5, gz, gy, gx] float32: flow x, y, z (post -> pre), peak sharpness, peak ratio."""
    pre = np.asarray(pre_image, dtype=np.float32)
    post = np.asarray(post_image, dtype=np.float32)
    grid = flow_grid_shape(pre.shape, patch_zyx, step_zyx)
    out = np.full((5,) + grid, np.nan, dtype=np.float32)
    for g in itertools.product(*[range(n) for n in grid]):
        sl = tuple(slice(i * s, i * s + p) for i, s, p in zip(g, step_zyx, patch_zyx))
        full = patch_xcorr_full(pre[sl], post[sl])
        pk, sharp, ratio = peak_stats(full, min_distance, peak_radius)
        d = [int(c) - (int(p) - 1) for c, p in zip(pk, patch_zyx)]
        out[(slice(None),) + g] = (d[2], d[1], d[0], sharp, ratio)
    return out


def clean_flow(flow, min_peak_ratio, min_peak_sharpness, max_magnitude, max_deviation, dim=3):
    """Invalidate (NaN) unreliable vectors; returns the first ``dim`` channels."""
    ret = np.array(flow[:dim], dtype=np.float32, copy=True)
    with np.errstate(invalid="ignore"):
        bad = np.abs(flow[dim]) < min_peak_sharpness
        pr = np.abs(flow[dim + 1])
        bad |= (pr > 0.0) & (pr < min_peak_ratio)
        ret[:, bad] = np.nan
        if max_magnitude > 0:
            ret[:, (np.abs(ret) > max_magnitude).any(axis=0)] = np.nan
        if max_deviation > 0:
            med = np.stack([sndi.median_filter(np.nan_to_num(c), size=3) for c in ret])
            ret[:, (np.abs(ret - med) > max_deviation).any(axis=0)] = np.nan
    return ret


def refine_flow_vectors_subpixel(flow_xyz, pre_image, post_image, patch_zyx, step_zyx, config):
    """sofima_registration.py:243-417 with scipy map_coordinates (order 1, constant 0)."""
    offsets = np.asarray(config.subpixel_offsets, dtype=np.float32)
    valid = np.argwhere(np.isfinite(flow_xyz[0]))
    if valid.size == 0:
        return flow_xyz.copy(), 0
    pre = np.asarray(pre_image, dtype=np.float32)
    post = np.asarray(post_image, dtype=np.float32)
    shape = np.asarray(pre.shape, dtype=np.float32)
    psz = np.asarray(patch_zyx, dtype=np.float32)
    vec_xyz = np.stack([flow_xyz[c][tuple(valid.T)] for c in range(3)], axis=1).astype(np.float32)
    post_starts = valid.astype(np.float32) * np.asarray(step_zyx, dtype=np.float32)
    base = post_starts + vec_xyz[:, ::-1]
    interior = np.all(base >= 1.0, axis=1) & np.all(base + psz <= shape - 2.0, axis=1)
    if not interior.any():
        return flow_xyz.copy(), 0
    valid, vec_xyz, post_starts, base = valid[interior], vec_xyz[interior], post_starts[interior], base[interior]
    cand = np.asarray(list(itertools.product(offsets, offsets, offsets)), dtype=np.float32)
    grid = np.indices(patch_zyx, dtype=np.float32)
    eps = np.float32(config.normalization_epsilon)
    refined = flow_xyz.copy()

    def _norm(p):
        p = p - p.mean(dtype=np.float32)
        return p / np.maximum(np.sqrt(np.mean(p * p, dtype=np.float32)), eps)

    for n in range(valid.shape[0]):
        pp = sndi.map_coordinates(post, grid + post_starts[n][:, None, None, None], order=1, mode="constant", cval=0.0)
        pp = _norm(pp.astype(np.float32))
        scores = np.empty(len(cand), dtype=np.float32)
        for c, off in enumerate(cand):
            q = sndi.map_coordinates(pre, grid + (base[n] + off)[:, None, None, None], order=1, mode="constant",
                                     cval=0.0)
            scores[c] = np.mean((_norm(q.astype(np.float32)) - pp) ** 2, dtype=np.float32)
        best = cand[int(np.argmin(scores))]
        new = vec_xyz[n] + best[::-1]
        for c in range(3):
            refined[c][tuple(valid[n])] = new[c]
    return refined.astype(np.float32, copy=False), int(valid.shape[0])


def median_initial_flow_field(cleaned):
    """sofima_registration.py:420-452."""
    init = np.zeros_like(cleaned, dtype=np.float32)
    ok = np.all(np.isfinite(cleaned), axis=0)
    if not ok.any():
        return init
    for c in range(3):
        init[c, ...] = float(np.median(cleaned[c][ok]))
        init[c][ok] = cleaned[c][ok]
    return init


def elastic_mesh_3d(x, k, stride_xyz):
    """Internal spring forces on the nodes; x [3 (xyz), z, y, x] relative node positions."""
    f = np.zeros_like(x)
    nz, ny, nx = x.shape[1:]
    for d in MESH_LINKS_XYZ:
        l0 = np.asarray(stride_xyz, dtype=np.float32) * np.asarray(d, dtype=np.float32)
        l0n = np.float32(np.sqrt(np.sum(l0 * l0)))
        kl = np.float32(k) * np.float32(stride_xyz[0]) / l0n
        sa, sb = [], []
        for dd, n in zip(d[::-1], (nz, ny, nx)):  # array axes z, y, x
            if dd == 0:
                sa.append(slice(0, n)); sb.append(slice(0, n))
            elif dd > 0:
                sa.append(slice(0, n - 1)); sb.append(slice(1, n))
            else:
                sa.append(slice(1, n)); sb.append(slice(0, n - 1))
        sa, sb = (slice(None),) + tuple(sa), (slice(None),) + tuple(sb)
        dx = x[sb] - x[sa] + l0[:, None, None, None]
        ln = np.sqrt(np.sum(dx * dx, axis=0, dtype=np.float32))
        fv = (kl * (np.float32(1.0) - l0n / ln)) * dx
        f[sa] += fv
        f[sb] -= fv
    return f


def stable_time_step(k0, k, stride_xyz):
    """Half the velocity-Verlet stability limit 2 / omega_max of the spring network, with the
    Gershgorin bound omega_max^2 <= k0 + 2 * (sum of the 18 link constants of an interior node)."""
    tot = 0.0
    for d in MESH_LINKS_XYZ:
        l0n = float(np.sqrt(sum((float(s) * c) ** 2 for s, c in zip(stride_xyz, d))))
        tot += float(k) * float(stride_xyz[0]) / l0n
    return 1.0 / float(np.sqrt(float(k0) + 4.0 * tot))


def relax_mesh(x0, prev, stride_xyz, config):
    """(relaxed x, kinetic energy per block, iterations)."""
    x = np.asarray(x0, dtype=np.float32).copy()
    prev = np.asarray(prev, dtype=np.float32)
    has_prev = np.isfinite(prev)
    prev0 = np.where(has_prev, prev, 0).astype(np.float32)
    k0, k = np.float32(config.mesh_k0), np.float32(config.mesh_k)
    cap = np.float32(config.mesh_start_cap)

    def force(x, cap):
        f = elastic_mesh_3d(x, k, stride_xyz) + np.where(has_prev, k0 * (prev0 - x), 0).astype(np.float32)
        return np.clip(f, -cap, cap)

    v = np.zeros_like(x)
    a = force(x, cap)
    dt, alpha, n_pos = np.float32(config.mesh_dt), np.float32(FIRE_ALPHA), 0
    # FIRE grows dt until the stiffest spring goes unstable; with force caps this large that throws
    # nodes across their neighbours (a folded mesh is another equilibrium), so dt is also bounded
    # by the integrator's stability limit
    dt_max = np.float32(min(float(config.mesh_dt_max), stable_time_step(k0, k, stride_xyz)))
    t, e_kin = 0, []
    while t < int(config.mesh_max_iters):
        for _ in range(int(config.mesh_num_iters)):
            x = x + dt * v + np.float32(0.5) * dt * dt * a
            a_new = force(x, cap)
            v = v + np.float32(0.5) * dt * (a + a_new)
            a = a_new
            a_norm = np.float32(np.sqrt(np.sum(a.astype(np.float64) ** 2) + 1e-6))
            v_norm = np.float32(np.sqrt(np.sum(v.astype(np.float64) ** 2)))
            power = float(np.sum(a.astype(np.float64) * v.astype(np.float64)))
            v = v + alpha * (a * (v_norm / a_norm) - v)
            n_pos = n_pos + 1 if power >= 0 else 0
            if power > 0 and n_pos > FIRE_N_MIN:
                dt = np.float32(min(dt * np.float32(FIRE_F_INC), dt_max))
                alpha = np.float32(alpha * np.float32(FIRE_F_ALPHA))
            if power < 0:
                dt = np.float32(dt * np.float32(FIRE_F_DEC))
                alpha = np.float32(FIRE_ALPHA)
                v = np.zeros_like(v)
        t += int(config.mesh_num_iters)
        e_kin.append(float(0.5 * np.sum(v.astype(np.float64) ** 2)))
        # a FIRE reset zeroes v for one step, so the residual force has to be small as well
        if max(float(np.max(np.abs(v))), float(np.max(np.abs(a)))) < float(config.mesh_stop_v_max):
            if cap >= np.float32(config.mesh_final_cap):
                break
            cap = np.float32(min(cap * np.float32(CAP_UPSCALE), np.float32(config.mesh_final_cap)))
    return x, e_kin, t


def stabilize_axial_flow_component(flow_xyz, config):
    """sofima_registration.py:80-146."""
    st = np.asarray(flow_xyz, dtype=np.float32).copy()
    axial = st[2]
    fin = np.isfinite(axial)
    if not fin.any():
        st[2] = 0.0
        return st, {"axial_flow_stabilized": True, "axial_flow_valid_vectors": 0, "axial_flow_median_px": 0.0,
                    "axial_flow_max_local_displacement_px": float(config.max_local_z_displacement_px),
                    "axial_flow_clipped_vectors": int(axial.size)}
    vals = axial[fin]
    med = float(np.median(vals))
    lo, hi = med - float(config.max_local_z_displacement_px), med + float(config.max_local_z_displacement_px)
    clipped = np.where(fin, np.clip(axial, lo, hi), med)
    n_clip = int(np.sum(np.abs(clipped - axial) > config.normalization_epsilon))
    st[2] = clipped.astype(np.float32, copy=False)
    return st, {"axial_flow_stabilized": True, "axial_flow_valid_vectors": int(fin.sum()),
                "axial_flow_median_px": med,
                "axial_flow_max_local_displacement_px": float(config.max_local_z_displacement_px),
                "axial_flow_clipped_vectors": n_clip, "axial_flow_preclip_min_px": float(vals.min()),
                "axial_flow_preclip_max_px": float(vals.max())}


def normalized_mean_squared_error(fixed, moving):
    """sofima_registration.py:149-175."""
    f = np.asarray(fixed, dtype=np.float32)
    m = np.asarray(moving, dtype=np.float32)
    eps = np.finfo(np.float32).eps
    fs, ms = max(float(np.std(f)), eps), max(float(np.std(m)), eps)
    fn = (f - float(np.mean(f))) / fs
    mn = (m - float(np.mean(m))) / ms
    return float(np.mean((fn - mn) ** 2))


def compose_flow_fields_same_grid(base, residual, stride_zyx, box_start_xyz):
    """sofima_registration.py:178-240 (the box start cancels: coords = grid + residual / stride)."""
    base = np.asarray(base, dtype=np.float32)
    residual = np.asarray(residual, dtype=np.float32)
    zg, yg, xg = np.indices(base.shape[1:], dtype=np.float32)
    shifted = (np.float32(box_start_xyz[2]) + zg * np.float32(stride_zyx[0]) + residual[2],
               np.float32(box_start_xyz[1]) + yg * np.float32(stride_zyx[1]) + residual[1],
               np.float32(box_start_xyz[0]) + xg * np.float32(stride_zyx[2]) + residual[0])
    coords = np.stack([(shifted[0] - np.float32(box_start_xyz[2])) / np.float32(stride_zyx[0]),
                       (shifted[1] - np.float32(box_start_xyz[1])) / np.float32(stride_zyx[1]),
                       (shifted[2] - np.float32(box_start_xyz[0])) / np.float32(stride_zyx[2])])
    sampled = np.stack([sndi.map_coordinates(base[c], coords, order=1, mode="nearest") for c in range(3)])
    return (residual + sampled).astype(np.float32)


def warp_with_flow(moving, flow_xyz, stride_zyx, box_start_xyz, reference_shape):
    """warp_array_to_reference_with_affine_and_sofima_flow_gpu (multiview_registration.py:905-1110)
    with an identity affine, unit spacing and mode="nearest" - the only way the estimator calls it
    (sofima_registration.py:589-598)."""
    moving = np.asarray(moving, dtype=np.float32)
    flow = np.asarray(flow_xyz, dtype=np.float32)
    zz, yy, xx = np.indices(tuple(int(v) for v in reference_shape), dtype=np.float32)
    fc = np.stack([(zz - np.float32(box_start_xyz[2])) / np.float32(stride_zyx[0]),
                   (yy - np.float32(box_start_xyz[1])) / np.float32(stride_zyx[1]),
                   (xx - np.float32(box_start_xyz[0])) / np.float32(stride_zyx[2])])
    dxyz = [sndi.map_coordinates(flow[c], fc, order=1, mode="nearest") for c in range(3)]
    coords = np.stack([zz + dxyz[2], yy + dxyz[1], xx + dxyz[0]])
    return sndi.map_coordinates(moving, coords, order=1, mode="nearest").astype(np.float32)


def _single_pass(fixed, moving, config):
    """sofima_registration.py:664-757."""
    if fixed.shape != moving.shape:
        raise ValueError("fixed_zyx and moving_affine_initialized_zyx must have matching shapes, "
                         f"got {fixed.shape!r} and {moving.shape!r}.")
    shape = tuple(int(v) for v in fixed.shape)
    patch, step = resolve_patch_and_step(shape, config)
    flow = patch_flow_field(moving, fixed, patch, step, int(config.peak_min_distance), int(config.peak_radius))
    cleaned = clean_flow(flow, float(config.min_peak_ratio), float(config.min_peak_sharpness),
                         float(config.max_magnitude), float(config.max_deviation), dim=3)
    cleaned, n_refined = refine_flow_vectors_subpixel(cleaned, moving, fixed, patch, step, config)
    n_valid = int(np.isfinite(cleaned[0]).sum())
    axial_md: dict[str, Any] = {}
    if n_valid == 0:
        field = np.zeros_like(cleaned, dtype=np.float32)
        status = "identity_fallback_no_valid_vectors"
        relax_md = {"mesh_relaxation": False, "mesh_iterations": 0, "mesh_final_kinetic_energy": 0.0}
    else:
        init = median_initial_flow_field(cleaned)
        field, e_kin, iters = relax_mesh(init, cleaned, (float(step[2]), float(step[1]), float(step[0])), config)
        relax_md = {"mesh_iterations": int(iters), "mesh_final_kinetic_energy": float(e_kin[-1]) if e_kin else 0.0,
                    "mesh_relaxation": True}
        field, axial_md = stabilize_axial_flow_component(field, config)
        status = "ok"
    stride = [float(v) for v in step]
    md = {"status": status, "valid_flow_vectors": n_valid, "subpixel_refined_vectors": n_refined,
          "residual_iterations": 1, "map_stride_zyx_px": stride,
          "map_box_start_xyz_px": [float(patch[2]) / 2.0, float(patch[1]) / 2.0, float(patch[0]) / 2.0],
          "map_box_size_xyz_px": [float((field.shape[3] - 1) * stride[2] + 1),
                                  float((field.shape[2] - 1) * stride[1] + 1),
                                  float((field.shape[1] - 1) * stride[0] + 1)],
          "mesh_initializer": "median_valid_flow", "sofima_config": config.as_metadata()}
    md.update(relax_md)
    md.update(axial_md)
    return field.astype(np.float32, copy=False), md


def estimate_sofima_flow_field_xyz_px(fixed, moving, config=None):
    """sofima_registration.py:524-662, 760-794."""
    config = config or SofimaRegistrationConfig()
    fixed = np.asarray(fixed, dtype=np.float32)
    moving = np.asarray(moving, dtype=np.float32)
    n_iter = int(config.residual_iterations)
    if n_iter <= 1:
        return _single_pass(fixed, moving, config)
    total, md = _single_pass(fixed, moving, config)

    def _warp(flow):
        return warp_with_flow(moving, flow, md["map_stride_zyx_px"], md["map_box_start_xyz_px"], fixed.shape)

    corrected = _warp(total)
    done = 1
    for it in range(1, n_iter):
        res, rmd = _single_pass(fixed, corrected, config)
        if rmd["status"] != "ok":
            break
        total = compose_flow_fields_same_grid(total, res, tuple(md["map_stride_zyx_px"]),
                                              tuple(md["map_box_start_xyz_px"]))
        total, amd = stabilize_axial_flow_component(total, config)
        md.update(amd)
        md["valid_flow_vectors"] = int(md["valid_flow_vectors"]) + int(rmd["valid_flow_vectors"])
        md["mesh_iterations"] = int(md["mesh_iterations"]) + int(rmd["mesh_iterations"])
        done += 1
        if it + 1 < n_iter:
            corrected = _warp(total)
    md["residual_iterations"] = done
    total, amd = stabilize_axial_flow_component(total, config)
    md.update(amd)
    corrected = _warp(total)
    affine_err = normalized_mean_squared_error(fixed, moving)
    sofima_err = normalized_mean_squared_error(fixed, corrected)
    md["affine_normalized_mse"] = affine_err
    md["sofima_normalized_mse"] = sofima_err
    if sofima_err >= affine_err:
        md["status"] = "identity_fallback_no_error_improvement"
        return np.zeros_like(total, dtype=np.float32), md
    return total.astype(np.float32, copy=False), md


__all__ = [n for n in dir() if not n.startswith("_")]
