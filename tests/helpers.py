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
