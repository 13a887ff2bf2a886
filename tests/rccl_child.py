"""Child process of tests/test_gpu_rccl_single_rank.py: joins an ``nccl`` (= RCCL) process group of ONE rank on
``cuda:0`` before any other GPU call and drives the code the N > 1 job runs - the device radix histogram
followed by ``dist.all_reduce`` inside ``PixelDecoder._global_normalization_vectors``
(reference: PixelDecoder.py:981-1199, tile sampling :1013-1020) and bench.py's timing collectives.
Prints one JSON line."""

import json
import os
import socket
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                      M3D_DIST_SINGLE_RANK="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist

    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)   # first GPU work of the process

    from conftest import pkg
    from helpers import codebook_frame, store_from_stack

    kernels, PD = pkg("kernels"), pkg("PixelDecoder")
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "dist_ready": bool(kernels._dist_ready())}

    # how many all-reduces of device int64[256] histograms really ran
    calls = {"n": 0, "cuda": 0, "shapes": set()}
    real_all_reduce = dist.all_reduce

    def counting_all_reduce(t, *a, **k):
        calls["n"] += 1
        calls["cuda"] += int(t.is_cuda)
        calls["shapes"].add((tuple(t.shape), str(t.dtype)))
        return real_all_reduce(t, *a, **k)

    dist.all_reduce = counting_all_reduce

    g = np.load(ROOT / "tests" / "golden" / "global_normalization.npz")
    readouts, predictors = g["readouts"], g["predictors"]
    n_bits = readouts.shape[1]
    bits = np.zeros((20, n_bits), dtype=np.int64)
    for c in range(20):
        bits[c, [(c + k * 3) % n_bits for k in range(4)]] = 1

    def vectors(single_rank_dist: bool):
        os.environ["M3D_DIST_SINGLE_RANK"] = "1" if single_rank_dist else "0"
        store = store_from_stack(list(readouts), codebook_frame(bits), predictors=list(predictors))
        dec = PD.PixelDecoder(store, merfish_bits=n_bits, verbose=0)
        dec._global_normalization_vectors(tile_indices=[0, 1, 2], lowpass_sigma=(3.0, 1.0, 1.0))
        norm, bkd = store.load_decode_normalization_vectors(None, "global")
        return np.asarray(norm, dtype=np.float32), np.asarray(bkd, dtype=np.float32)

    norm_d, bkd_d = vectors(True)
    n_dist_calls = calls["n"]
    norm_l, bkd_l = vectors(False)          # group=False route: no collective
    out["collective_calls"] = n_dist_calls
    out["collective_calls_local_route"] = calls["n"] - n_dist_calls
    out["collective_on_device"] = calls["cuda"] == calls["n"]
    out["collective_shapes"] = sorted(str(s) for s in calls["shapes"])
    out["norm_bits_equal"] = bool(np.array_equal(norm_d.view(np.uint32), norm_l.view(np.uint32)))
    out["bkd_bits_equal"] = bool(np.array_equal(bkd_d.view(np.uint32), bkd_l.view(np.uint32)))
    out["norm_vs_reference_run"] = float(np.abs(norm_d / g["normalization"] - 1).max())
    out["bkd_vs_reference_run"] = float(np.abs(bkd_d / g["background"] - 1).max())

    # a pooled statistic straight through PooledValues (group=None = the default group)
    os.environ["M3D_DIST_SINGLE_RANK"] = "1"
    rng = np.random.default_rng(7)
    a = rng.gamma(2.0, 30.0, size=200_003).astype(np.float32)
    pool = kernels.PooledValues()
    pool.add(torch.from_numpy(a).to(device), sub=4.0, above=50.0)
    ref = np.maximum(a - np.float32(4.0), 0)
    ref = ref[ref > 50.0]
    out["pool_count_ok"] = pool.count() == int(ref.size)
    out["pool_median_ok"] = pool.median() == float(np.median(ref))

    dist.all_reduce = real_all_reduce
    sys.argv = ["bench.py"]
    import bench

    out["aggregate"] = list(bench.aggregate_over_ranks(1.25, 37.0, 1, device))
    out["agree_min"] = bench.agree_min(11, 1, device)
    dist.barrier()
    torch.cuda.synchronize(device)
    dist.destroy_process_group()
    print("RCCL_CHILD " + json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
