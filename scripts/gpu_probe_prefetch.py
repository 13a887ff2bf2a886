#!/usr/bin/env python3
"""decode_all_tiles from an on-disk OME-Zarr store: with and without the next tile's read-ahead
(ZarrImageDataStore.prefetch_tile), against the same tiles held in memory.
Usage: gpu_probe_prefetch.py [tiles Z Y X bits]"""
import importlib
import json
import shutil
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

DS = importlib.import_module("merfish3d-analysis_amd.datastore")
PD = importlib.import_module("merfish3d-analysis_amd.PixelDecoder")
kernels = importlib.import_module("merfish3d-analysis_amd.kernels")

args = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else [4, 32, 1024, 1024, 16]
n_tiles, shape, n_bits = args[0], tuple(args[1:4]), args[4]
dev = torch.device("cuda:0")
codebook = bench.hw4_codebook(n_bits)
bits = codebook.iloc[:, 1:].to_numpy(dtype=np.int64)
root = Path(tempfile.mkdtemp(prefix="m3d_prefetch_", dir="/dev/shm" if Path("/dev/shm").exists() else None))
kw = dict(num_tiles=n_tiles, num_rounds=1, num_bits=n_bits, codebook=codebook)
mem = DS.InMemoryDataStore(**kw)
disk = DS.ZarrImageDataStore(root, device="cuda:0", **kw)
n_em = int(80_000 * np.prod(shape) / (64 * 2048 * 2048))
for t in range(n_tiles):
    ro, pr, _, _ = bench.make_tile(shape, bits, n_em, 10 + t, dev, kernels)
    for b in range(n_bits):
        for store in (mem, disk):
            store.set_corrected_image(ro[b], tile=t, bit=b)
            store.save_local_feature_predictor_image(pr[b], tile=t, bit=b)
    for store in (mem, disk):
        store.set_stage_position_zyx_um(t, (0.0, t * shape[1] * 0.098, 0.0))
    print(f"tile {t} stored", flush=True)
bkd, nrm = np.full(n_bits, 1.0, np.float32), np.full(n_bits, 300.0, np.float32)
for store in (mem, disk):
    store.save_decode_normalization_vectors(None, "global", nrm, bkd)


def run(store, label, prefetch=True):
    dec = PD.PixelDecoder(store, merfish_bits=n_bits, verbose=0)
    saved = getattr(store, "prefetch_tile", None)
    if saved is not None and not prefetch:
        store.prefetch_tile = lambda *a, **k: 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dec.decode_all_tiles(assign_to_cells=False, normalization_method="global", minimum_pixels=3)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if saved is not None and not prefetch:
        del store.prefetch_tile
    table = store.load_global_filtered_decoded_spots()
    print(label, round(dt, 3), "s", len(table), "rows", flush=True)
    return dt, table


out = {"tiles": n_tiles, "shape": shape, "bits": n_bits}
run(mem, "warm-up (memory)")
out["memory_s"], ref = run(mem, "memory")
out["disk_no_prefetch_s"], a = run(disk, "disk, no read-ahead", prefetch=False)
out["disk_prefetch_s"], b = run(disk, "disk, read-ahead", prefetch=True)
cols = ["gene_id", "tile_idx", "global_z", "global_y", "global_x"]
key = lambda df: df[cols].sort_values(cols).reset_index(drop=True)
out["tables_equal"] = bool(key(ref).equals(key(a)) and key(ref).equals(key(b)))
out["rows"] = len(ref)
shutil.rmtree(root, ignore_errors=True)
print(json.dumps(out))
