"""Test infrastructure: oracle jobs that run in spawned worker processes (no torch, no GPU in the children), so that a
GPU test can have sixteen NumPy deconvolutions run side by side on the box's host cores."""

from __future__ import annotations

import warnings


def rlgc_to_convergence(job):
    """(image uint16, psf, seed, fft_workers[, max_iterations]) -> (float32 volume, iterations, stop reason, KL traces
    of split 1 and split 2)."""
    import sys
    from pathlib import Path

    root = str(Path(__file__).resolve().parents[1])
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import rlgc_ref

    image, psf, seed, workers = job[:4]
    cap = job[4] if len(job) > 4 else None
    rlgc_ref.FFT_WORKERS = int(workers)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out, info = rlgc_ref.rlgc(image, psf, rng="philox", rng_seed=seed, return_info=True, max_iterations=cap)
    return out, info.iterations, info.stop_reason, list(info.kld_split1), list(info.kld_split2)


def rlgc_reference_stream(job):
    """(image uint16, psf, seed, fft_workers) -> (float32 volume, iterations, stop reason, splits uint16 [iterations + 1,
    Z, Y, X]): the oracle with NumPy's binomial stream - the stream the reference-run fixtures were made with, and
    bit for bit the reference's own run (tests/test_oracle_golden.py) - recording the draws so that the HIP solver can
    be handed the same ones."""
    import sys
    from pathlib import Path

    import numpy as np

    root = str(Path(__file__).resolve().parents[1])
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import rlgc_ref

    image, psf, seed, workers = job
    rlgc_ref.FFT_WORKERS = int(workers)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out, info = rlgc_ref.rlgc(image, psf, rng="numpy", rng_seed=seed, return_info=True, record_splits=True)
    splits = np.stack([rlgc_ref.remove_padding_zyx(s, info.pad_width) for s in info.splits]).astype(np.uint16)
    return out, info.iterations, info.stop_reason, splits
