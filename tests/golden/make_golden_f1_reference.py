#!/usr/bin/env python3
"""One cell of the simulation F1 matrix run through the REFERENCE's own code (authoring container only):

    python tests/golden/make_golden_f1_reference.py [cell ...]     # ~15 min per cell on 8 cores; default: dz0.315

Dataset: a cell of the simulation matrix of tests/oracle_pipeline.py (dz0.315 / dz1.0 / dz1.5: 16 x 256 x 256, 8 rounds x
(fiducial + 2 bits), seeds fixed; the two coarser spacings decode per plane, PixelDecoder's 2-D mode).  With the stand-ins of make_golden.py (cupy -> NumPy, cupyx.scipy.ndimage -> SciPy, cuVS distance -> expanded
float32 form, scikit-image / cuCIM label + regionprops -> oracle.ccl_ref, itself pinned to scikit-image 0.18.3) the
reference executes, chained on ONE dataset for the first time:

  utils/rlgc.py ``chunked_rlgc``                    every fiducial and readout volume (DataRegistration.py:379-387 call form,
                                                    clip -> uint16 as :435, :993-995)
  PixelDecoder.optimize_normalization_by_decoding   n_random_tiles=1, n_iterations=3 (the simulation CLI's setting)
  PixelDecoder.decode_all_tiles                     decode_one_tile / _extract_barcodes per tile, blank-fraction filter,
                                                    save of the filtered table (PixelDecoder.py:4759-4868)
  cli/statphysbio_simulation/calculate_F1.py        ``calculate_F1`` itself (coordinate conventions + calculate_F1_with_radius)

What is NOT the reference's code: the round transforms (``register_pair_to_fixed`` needs cuCIM's phase correlation,
absent here: both runs decode with the PLANTED shifts as translations, stored in the fixture - this pins the affine
column downstream of registration; what the oracle's restatement finds is recorded for information), the predictor images (U-FISH, out of scope:
the shared difference-of-Gaussians stand-in of tests/oracle_pipeline.py) and the transforms' FFT backend (scipy.fft
with worker threads stands in for cuFFT where make_golden.py uses numpy.fft; same mathematics, last-bit differences).

Output: tests/golden/f1_reference_run.json, one entry per cell - the scorer's dictionary, the round transforms,
per-volume RLGC iteration counts (from the reference's own log lines), transcript counts before / after the filter, the
iterative normalisation vectors, and ``oracle_pipeline_same_transforms``: the ORACLE pipeline (Philox split stream) run
from the same transforms, the twin the HIP pipeline's own stream is held to.  tests/test_gpu_f1_reference_run.py runs the HIP pipeline from the same transforms and asserts
|F1_hip - F1_reference| <= 0.005 (BASELINE.json north_star); tests/test_oracle_golden.py holds the oracle pipeline's
affine cell (f1_matrix.json) to the same bound.
"""

from __future__ import annotations

import json
import logging
import re
import sys
import tempfile
import time
from pathlib import Path
from types import SimpleNamespace

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import make_golden as mg  # noqa: E402  (stand-ins + import of the reference)

def run_cell(CELL: str) -> dict:
    t_start = time.time()

    def progress(msg):
        print(f"[f1-reference +{time.time() - t_start:6.0f}s] {msg}", flush=True)

    rl, _mv, _dw, _sf, pdm = mg.import_reference()
    import scipy.fft as sfft

    import cupy as cp  # the stand-in module

    cp.fft.rfftn = lambda a, *x, **k: sfft.rfftn(a, *x, workers=8, **k)
    cp.fft.irfftn = lambda a, *x, **k: sfft.irfftn(a, *x, workers=8, **k)

    import oracle_pipeline as op
    from oracle import build as oracle_build
    from oracle import registration_ref

    oracle_build.build()
    iters = {}

    class _Grab(logging.Handler):
        def emit(self, record):
            m = re.search(r"solver_completed.*?iterations=(\d+)", record.getMessage())
            if m:
                iters.setdefault(self.key, []).append(int(m.group(1)))

    grab = _Grab()
    grab.key = None
    logger = logging.getLogger("m3d.f1.reference.rlgc")
    logger.setLevel(logging.INFO)
    logger.addHandler(grab)

    with tempfile.TemporaryDirectory(prefix="m3d_f1_reference_") as work:
        work = Path(work)
        store, gt = op.make_cell(work, CELL)
        cell = op.CELLS[CELL]
        t = store.tile_ids[0]
        spacing = tuple(float(v) for v in store.voxel_size_zyx_um)
        psfs = store.channel_psfs

        def decon_u16(image, psf, key):
            grab.key = key
            out = rl.chunked_rlgc(image=np.asarray(image), psf=np.asarray(psf, dtype=np.float32), gpu_id=0, crop_yx=2048,
                                  crop_z=None, release_memory=False, logger=logger)
            return np.clip(out, 0, 65535).astype(np.uint16)          # DataRegistration.py:435, 993-995

        fid = []
        for rid in store.round_ids:
            raw = store.load_local_corrected_image(tile=t, round=rid, return_future=False)
            fid.append(decon_u16(raw, psfs[0], f"fiducial/{rid}"))
            store.save_local_deconvolved_fiducial_image(fid[-1], tile=t, round=rid)
            progress(f"reference chunked_rlgc fiducial {rid}: {iters.get(f'fiducial/{rid}')}")
        # Round transforms: the PLANTED shifts as pure translations (pull offset = + shift, the generator's convention),
        # identical for this run and for the HIP run of tests/test_gpu_f1_reference_run.py.  What the oracle's
        # restatement of register_pair_to_fixed finds on these reference-deconvolved fiducials is recorded beside them:
        # on this dataset it registers round 6 an image width off - scikit-image's disambiguation rule accepting a
        # three-sample corner tile - which is why the comparison does not ride on it.
        sp = np.asarray(spacing, dtype=np.float64)
        transforms, found = [], []
        for r, rid in enumerate(store.round_ids):
            T = np.eye(4, dtype=np.float32)
            T[:3, 3] = (np.asarray(cell["shifts"][r], dtype=np.float64) * sp).astype(np.float32)
            transforms.append(T)
            store.save_local_round_transform_zyx_um(T, tile=t, round=rid)
            if r == 0:
                found.append([0.0, 0.0, 0.0])
            else:
                F = np.asarray(registration_ref.register_pair_to_fixed(fid[0], fid[r], spacing), dtype=np.float64)
                found.append((F[:3, 3] / sp).round(3).tolist())
        progress(f"planted shifts {[list(s_) for s_ in cell['shifts']]}; oracle restatement finds {found}")
        for bit_id in store.bit_ids:
            ex, _em = store.load_local_wavelengths_um(tile=t, bit=bit_id)
            raw = store.load_local_corrected_image(tile=t, bit=bit_id, return_future=False)
            decon = decon_u16(raw, psfs[1 if ex < 0.600 else 2], f"readout/{bit_id}")
            store.save_local_deconvolved_readout_image(decon, tile=t, bit=bit_id)
            _spots, prob = op.predictor(decon)
            store.save_local_feature_predictor_image(prob, tile=t, bit=bit_id)
            progress(f"reference chunked_rlgc readout {bit_id}: {iters.get(f'readout/{bit_id}')}")

        # ---- the reference's decoder on this store (technique of make_golden.make_optimize)
        label, rso, rpt = mg._skimage_standins()
        pdm.label, pdm.remove_small_objects = label, rso
        pdm.gpu_regionprops_table, pdm.regionprops_table = rpt, rpt
        pdm._start_gpu_worker_process = lambda target, args, physical_gpu_id: target(*args)
        pdm._join_gpu_workers = lambda procs, what: None
        pdm.preload_cuda_libraries = lambda: None
        pdm.qi2labDataStore = lambda path, validate=False: store
        real_torch = sys.modules.get("torch")
        sys.modules["torch"] = SimpleNamespace(cuda=SimpleNamespace(set_device=lambda i: None))
        try:
            dec = pdm.PixelDecoder(store, merfish_bits=16, num_gpus=1, verbose=0)
            kw = dict(cell["decode"])
            dec.optimize_normalization_by_decoding(n_random_tiles=1, n_iterations=3, **kw)
            progress("reference optimize_normalization_by_decoding done")
            dec.decode_all_tiles(assign_to_cells=False, normalization_method="iterative", **kw)
            progress("reference decode_all_tiles done")
        finally:
            if real_torch is not None:
                sys.modules["torch"] = real_torch
            else:
                sys.modules.pop("torch", None)

        # ---- the reference's scorer CLI function, pointed at this store
        import importlib

        f1mod = importlib.import_module("merfish3danalysis.cli.statphysbio_simulation.calculate_F1")
        f1mod.qi2labDataStore = lambda path: store
        root = work / CELL                                         # holds GT_spots.csv (simulation.make_simulation)
        results = f1mod.calculate_F1(root, search_radius=float(cell["radius"]))
        norm, bkd = store.load_decode_normalization_vectors(None, "iterative")
        out = {
            "cell": CELL, "radius": cell["radius"], "decode": {k: (list(v) if isinstance(v, tuple) else v) for k, v in cell["decode"].items()},
            "seed": cell["seed"], "shifts": [list(s) for s in cell["shifts"]],
            "f1": {k: (float(v) if isinstance(v, (float, np.floating)) else int(v)) for k, v in results.items()},
            "round_transforms_zyx_um": [np.asarray(T, dtype=np.float64).round(6).tolist() for T in transforms],
            "shifts_found_by_oracle_registration_px": found,
            "rlgc_iterations": iters,
            "decoded_before_filter": int(len(dec._df_barcodes_loaded)),
            "decoded_after_filter": int(len(dec._df_filtered_barcodes)),
            "normalization": np.round(np.asarray(norm, dtype=float), 1).tolist(),
            "background": np.round(np.asarray(bkd, dtype=float), 1).tolist(),
            "seconds": round(time.time() - t_start, 1),
        }
        # ---- the oracle pipeline (the build's Philox stream) from the same transforms, on a fresh copy of the dataset
        t1 = time.time()
        store2, gt2 = op.make_cell(work / "oracle", CELL)
        res = op.oracle_readouts_then_score(store2, gt2, transforms, float(cell["radius"]), dict(cell["decode"]))
        out["oracle_pipeline_same_transforms"] = {
            k: (float(v) if isinstance(v, (float, np.floating)) else (int(v) if isinstance(v, (int, np.integer)) else v))
            for k, v in res.items()}
        out["oracle_pipeline_seconds"] = round(time.time() - t1, 1)
        progress(f"oracle pipeline, same transforms: {out['oracle_pipeline_same_transforms']['F1 Score']:.4f}")
    return out


def main() -> None:
    cells = sys.argv[1:] or ["dz0.315"]
    path = HERE / "f1_reference_run.json"
    results = json.loads(path.read_text()) if path.exists() else {}
    for name in cells:
        results[name] = run_cell(name)
        path.write_text(json.dumps(results, indent=1, sort_keys=True))
        print("wrote", path, name, results[name]["f1"])


if __name__ == "__main__":
    main()
