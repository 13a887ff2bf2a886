"""Scratch probe: where decode_one_tile spends host time on an on-disk qi2lab store (metadata reads, table write)
against the in-memory store - cProfile of one call each on a small tile.  usage: gpu_probe_decode_store.py"""
import cProfile
import importlib
import pstats
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.argv = ["bench.py"]
import bench  # noqa: E402

DS = importlib.import_module("merfish3d-analysis_amd.datastore")
QS = importlib.import_module("merfish3d-analysis_amd.qi2lab_store")
PD = importlib.import_module("merfish3d-analysis_amd.PixelDecoder")
kernels = importlib.import_module("merfish3d-analysis_amd.kernels")
dev = torch.device("cuda:0")
shape, n_bits = (16, 512, 512), 22
cb = bench.hw4_codebook(n_bits)
mat = cb.iloc[:, 1:].to_numpy(dtype=np.int64)
readouts, predictors, truth, _f = bench.make_tile(shape, mat, 5000, 3, dev, kernels)
with tempfile.TemporaryDirectory(dir="/dev/shm") as tmp:
    stores = {"memory": DS.InMemoryDataStore(num_tiles=1, num_rounds=11, num_bits=n_bits, codebook=cb),
              "disk": QS.Qi2labDiskStore(Path(tmp) / "ds", device=dev, num_threads=4, write_behind=True, num_tiles=1,
                                         num_rounds=11, num_bits=n_bits, codebook=cb)}
    for name, store in stores.items():
        for b in range(n_bits):
            store.set_corrected_image(readouts[b], tile=0, bit=b)
            store.save_local_deconvolved_readout_image(readouts[b], tile=0, bit=b)
            store.save_local_feature_predictor_image(predictors[b], tile=0, bit=b)
            store.set_round_linker(store.tile_ids[0], store.bit_ids[b], 1 + b // 2)
        for r in range(11):
            T = np.eye(4, dtype=np.float32)
            T[:3, 3] = (0.0, 0.1 * r, -0.05 * r)
            store.save_local_round_transform_zyx_um(T, tile=0, round=store.round_ids[r])
        store.save_decode_normalization_vectors(None, "global", np.full(n_bits, 300, np.float32), np.full(n_bits, 20, np.float32))
        dec = PD.PixelDecoder(store, merfish_bits=n_bits, verbose=0)
        dec.decode_one_tile(tile_idx=0, normalization_method="global")   # warm
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        dec.decode_one_tile(tile_idx=0, normalization_method="global")
        store.save_local_decoded_spots(dec.decoded_barcodes, tile=0)
        torch.cuda.synchronize()
        pr.disable()
        print(f"==== {name}: {len(dec.decoded_barcodes)} transcripts")
        pstats.Stats(pr).sort_stats("cumtime").print_stats(22)
