"""Shared builders for the operator tests (in-memory datastore from golden fixtures)."""

from __future__ import annotations

import numpy as np
import pandas as pd

from conftest import pkg


def codebook_frame(bits: np.ndarray, gene_ids=None) -> pd.DataFrame:
    n_codes, n_bits = bits.shape
    if gene_ids is None:
        gene_ids = [f"gene{i:03d}" for i in range(n_codes)]
    df = pd.DataFrame(np.asarray(bits, dtype=np.int64), columns=[f"bit{i + 1:02d}" for i in range(n_bits)])
    df.insert(0, "gene_id", [str(g) for g in gene_ids])
    return df


def store_from_stack(stack_list, codebook_df, voxel=(0.315, 0.098, 0.098), microscope_type="3D",
                     predictors=None, to_device=None):
    """One tile per entry of ``stack_list`` ([B,Z,Y,X] arrays); identity registration."""
    ds_mod = pkg("datastore")
    n_bits = stack_list[0].shape[0]
    store = ds_mod.InMemoryDataStore(num_tiles=len(stack_list), num_rounds=1, num_bits=n_bits,
                                     codebook=codebook_df, voxel_size_zyx_um=voxel,
                                     microscope_type=microscope_type)
    for t, stack in enumerate(stack_list):
        for b in range(n_bits):
            img = stack[b]
            pred = np.ones(img.shape, dtype=np.float32) if predictors is None else predictors[t][b]
            if to_device is not None:
                import torch

                img = torch.from_numpy(np.ascontiguousarray(img)).to(to_device)
                pred = torch.from_numpy(np.ascontiguousarray(pred)).to(to_device)
            store.set_corrected_image(img, tile=t, bit=b)
            store.save_local_feature_predictor_image(pred, tile=t, bit=b)
    return store


def sort_table(df: pd.DataFrame) -> pd.DataFrame:
    cols = ["tile_z", "tile_y", "tile_x", "gene_id", "area"]
    return df.sort_values(cols, kind="stable").reset_index(drop=True)


def use_scipy_lowpass(monkeypatch):
    """Replace the package's Gaussian low-pass by SciPy's (float64 accumulation) for one test: that is what the
    reference run behind the fixtures used (cupyx.scipy.ndimage stand-in), so everything downstream of the low-pass
    can be asserted exactly instead of to the float32-vs-float64 rounding of the filter.  The HIP low-pass itself is
    compared with SciPy in tests/test_gpu_decode.py."""
    import scipy.ndimage as sndi
    import torch

    kernels = pkg("kernels")

    def lowpass(volume, sigma_zyx, is_3d=True, out=None, scratch=None):
        img = volume.detach().cpu().numpy().astype(np.float32)
        if is_3d:
            res = sndi.gaussian_filter(img, sigma=tuple(float(v) for v in sigma_zyx))
        else:
            res = np.stack([sndi.gaussian_filter(p, sigma=(float(sigma_zyx[1]), float(sigma_zyx[2]))) for p in img])
        t = torch.from_numpy(np.ascontiguousarray(res, dtype=np.float32)).to(volume.device)
        if out is not None:
            out.copy_(t)
            return out
        return t

    monkeypatch.setattr(kernels, "gaussian_lowpass", lowpass)
