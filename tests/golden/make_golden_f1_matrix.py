#!/usr/bin/env python3
"""F1 of the ORACLE pipeline on the simulation matrix (tests/oracle_pipeline.py): three axial spacings (0.315 / 1.0 /
1.5 um, F1 radii 1.0 / 1.0 / 1.5 um; tests/test_simulation_example_pipeline.py:48-52) x {affine only, affine + SOFIMA
residual flow}, datasets of 16 x 256 x 256 written by ``simulation.make_simulation``.  CPU only, ~1 h on 8 cores
(NumPy RLGC of 24 volumes per cell dominates):

    python tests/golden/make_golden_f1_matrix.py [cell ...]

Output: tests/golden/f1_matrix.json - per cell and mode the scorer's dictionary (F1, precision, recall, TP / FP / FN),
the registration transforms the oracle found and the SOFIMA status per round.  tests/test_gpu_f1_matrix.py regenerates
the same datasets from their seeds on the GPU box, runs the HIP pipeline and asserts |F1_hip - F1_oracle| <= 0.005
(BASELINE.json north_star) and F1(SOFIMA) >= F1(affine) to three decimals.  Nothing of the reference is executed
here: its F1 scorer and post-decode filters are pinned separately (postdecode.npz)."""

from __future__ import annotations

import copy
import json
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import oracle_pipeline as op  # noqa: E402
from oracle import build as oracle_build  # noqa: E402


def run_cell(name: str, work: Path) -> dict:
    t0 = time.time()

    def progress(msg):
        print(f"[{name} +{time.time() - t0:6.0f}s] {msg}", flush=True)

    base, gt = op.make_cell(work, name)
    out = {}
    # the deformable run repeats the affine run's deconvolutions and registrations bit for bit (same seeds): do them
    # once, then branch
    op.oracle_preprocess(base, deformable=False, progress=progress)
    for mode in ("affine", "sofima"):
        store = copy.deepcopy(base)
        if mode == "sofima":
            t = store.tile_ids[0]
            spacing = tuple(float(v) for v in store.voxel_size_zyx_um)
            fixed = np.asarray(store.load_local_fiducial_image(tile=t, round=store.round_ids[0], return_future=False),
                               dtype=np.float32)
            status = {}
            for rid in store.round_ids[1:]:
                moving = np.asarray(store.load_local_fiducial_image(tile=t, round=rid, return_future=False))
                T = store.load_local_round_transform_zyx_um(tile=t, round=rid)
                warped = op.registration_ref.warp_array_to_reference(moving, T, spacing, fixed.shape)
                flow, md = op.sofima_ref.estimate_sofima_flow_field_xyz_px(fixed, warped.astype(np.float32))
                store.save_local_sofima_flow_field(
                    flow, tile=t, round=rid, reference_round=store.round_ids[0],
                    map_stride_zyx_px=md["map_stride_zyx_px"], map_box_start_xyz_px=md["map_box_start_xyz_px"],
                    map_box_size_xyz_px=md["map_box_size_xyz_px"], reference_shape_zyx_px=tuple(fixed.shape),
                    moving_shape_zyx_px=tuple(warped.shape), sofima_status=md.get("status", "ok"),
                    valid_flow_vectors=md.get("valid_flow_vectors"), return_future=False)
                status[rid] = [md.get("status"), int(md.get("valid_flow_vectors") or 0)]
                progress(f"sofima {rid}: {status[rid]}")
            out["sofima_status"] = status
        res = op.oracle_decode_and_score(store, gt, op.CELLS[name]["radius"], op.CELLS[name]["decode"])
        out[mode] = {k: (float(v) if isinstance(v, (float, np.floating)) else (int(v) if isinstance(v, (int, np.integer))
                                                                                  else v)) for k, v in res.items()}
        progress(f"{mode}: {out[mode]}")
        if mode == "affine":
            t = store.tile_ids[0]
            sp = np.asarray(store.voxel_size_zyx_um)
            out["shifts_found_px"] = [(np.asarray(store.load_local_round_transform_zyx_um(t, r))[:3, 3] / sp).round(3).tolist()
                                      for r in store.round_ids]
    out["seconds"] = round(time.time() - t0, 1)
    return out


def main() -> None:
    oracle_build.build()
    cells = sys.argv[1:] or list(op.CELLS)
    path = HERE / "f1_matrix.json"
    results = json.loads(path.read_text()) if path.exists() else {}
    with tempfile.TemporaryDirectory(prefix="m3d_f1_matrix_") as work:
        for name in cells:
            results[name] = run_cell(name, Path(work))
            results["_meta"] = {"shape": list(op.SHAPE),
                                "cells": {k: {"dz": c["dz"], "radius": c["radius"], "seed": c["seed"],
                                              "decode": {a: (list(b) if isinstance(b, tuple) else b)
                                                         for a, b in c["decode"].items()},
                                              "shifts": [list(s) for s in c["shifts"]]} for k, c in op.CELLS.items()}}
            path.write_text(json.dumps(results, indent=1, sort_keys=True))
    print("wrote", path)


if __name__ == "__main__":
    main()
