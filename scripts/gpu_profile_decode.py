#!/usr/bin/env python3
"""Host-side profile (cProfile) of PixelDecoder.decode_one_tile on a config-2 sized synthetic tile."""
import cProfile
import importlib
import pstats
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

kernels = importlib.import_module("merfish3d-analysis_amd.kernels")
PD = importlib.import_module("merfish3d-analysis_amd.PixelDecoder")
DS = importlib.import_module("merfish3d-analysis_amd.datastore")
shape, n_bits = (64, 2048, 2048), 22
dev = torch.device("cuda:0")
codebook = bench.hw4_codebook(n_bits)
bits = codebook.iloc[:, 1:].to_numpy(dtype=np.int64)
ro, pr, _, _ = bench.make_tile(shape, bits, 80_000, 1, dev, kernels)
store = DS.InMemoryDataStore(num_tiles=1, num_rounds=1, num_bits=n_bits, codebook=codebook)
for b in range(n_bits):
    store.set_corrected_image(ro[b], tile=0, bit=b)
    store.save_local_feature_predictor_image(pr[b], tile=0, bit=b)
dec = PD.PixelDecoder(store, merfish_bits=n_bits, verbose=0)
dec._tile_idx = 0
dec._load_bit_data(gpu_id=0)
dec._lp_filter(sigma=PD.DEFAULT_DECODE_LOWPASS_SIGMA, gpu_id=0)
bkd, nrm = np.zeros(n_bits, np.float32), np.ones(n_bits, np.float32)
for b in range(n_bits):
    pool = kernels.PooledValues(group=False)
    pool.add(dec._image_data_lp[b])
    bkd[b] = pool.percentile(50.0)
    nrm[b] = max(pool.percentile(99.9) - bkd[b], 1e-3)
store.save_decode_normalization_vectors(None, "global", nrm, bkd)
dec._cleanup()
dec.decode_one_tile(tile_idx=0, gpu_id=0, normalization_method="global")   # warm
torch.cuda.synchronize()
t0 = time.perf_counter()
pr_ = cProfile.Profile()
pr_.enable()
dec.decode_one_tile(tile_idx=0, gpu_id=0, normalization_method="global")
torch.cuda.synchronize()
pr_.disable()
print("decode_one_tile", round(time.perf_counter() - t0, 3), "s;", len(dec.decoded_barcodes), "rows")
pstats.Stats(pr_).sort_stats("cumulative").print_stats(45)
