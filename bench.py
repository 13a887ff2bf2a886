#!/usr/bin/env python3
"""Headline benchmark: Mvoxels/s through decon + decode on synthetic MERFISH tiles.

One *step* = the per-tile hot path on one synthetic tile held in HBM:
B x RLGC deconvolution of the readout images (one per bit, clip -> uint16 like
DataRegistration.py:993-995) followed by one full ``PixelDecoder.decode_one_tile``
(readout x predictor, low-pass, fused pixel decode, connected components, transcript
table).  Default workload = BASELINE.json ``configs[1]``: 64 x 2048 x 2048 tile, 22-bit /
140-codeword codebook, PSF 64 x 51 x 51 (FFT volume 128 x 2187 x 2187).

    python bench.py --gpus 1 --steps 1 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Tiles are independent: every rank decodes its own tile per step (weak scaling, no
data-path collective; RCCL is only used for the timing barrier / max-reduce here).
Rank 0 prints ONE JSON line.  ``roofline`` prices one RLGC iteration (the unit that is launched
~1600 times per tile) at SURVEY.md section 8(d)'s 148 algorithmic bytes per padded voxel over the
iteration's device time measured with HIP events inside ``m3d_rlgc_run``; ``roofline_conv`` and
``roofline_kernels`` break it down, ``roofline_decode`` is the MFMA-bound decode kernel.
"""

from __future__ import annotations

import argparse
import importlib
import itertools
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import pandas as pd
import torch

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

WORKLOADS = {
    # name: (tile zyx, bits, psf zyx, emitters)
    "config2": ((64, 2048, 2048), 22, (64, 51, 51), 80_000),
    "config1": ((16, 256, 256), 16, (16, 51, 51), 1_200),
}
RLGC_ALGORITHMIC_BYTES_PER_ITER_VOXEL = 148.0
RLGC_ALGORITHMIC_BYTES_PER_CONV_VOXEL = 20.0  # 2 FFTs x 8 B + OTF read 4 B
F32_MFMA_PEAK_TFLOPS = 157.3  # dense fp32 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def hw4_codebook(n_bits: int, n_codes: int = 140, n_blank: int = 10, seed: int = 0) -> pd.DataFrame:
    """Weight-4 / Hamming-distance-4 codebook (MHD4 family used by the reference data).  16 bits: the 140
    blocks of the Steiner quadruple system on GF(2)^4 (4-subsets XOR-ing to zero, the optimum A(16,4,4));
    other lengths: a seeded greedy selection."""
    rng = np.random.default_rng(seed)
    combos = list(itertools.combinations(range(n_bits), 4))
    chosen: list[tuple] = []
    if n_bits == 16:
        chosen = [c for c in combos if c[0] ^ c[1] ^ c[2] ^ c[3] == 0][:n_codes]
    else:
        rng.shuffle(combos)
        for c in combos:
            s = set(c)
            if all(len(s & set(o)) <= 2 for o in chosen):
                chosen.append(c)
            if len(chosen) == n_codes:
                break
    bits = np.zeros((len(chosen), n_bits), dtype=np.int64)
    for i, c in enumerate(chosen):
        bits[i, list(c)] = 1
    names = [f"gene{i:03d}" for i in range(len(chosen) - n_blank)] + [f"Blank-{i:02d}" for i in range(n_blank)]
    df = pd.DataFrame(bits, columns=[f"bit{i + 1:02d}" for i in range(n_bits)])
    df.insert(0, "gene_id", names)
    return df


def gaussian_psf(shape, sigma, device):
    zz, yy, xx = torch.meshgrid(*[torch.arange(n, device=device, dtype=torch.float64) - n // 2 for n in shape],
                                indexing="ij")
    psf = torch.exp(-0.5 * ((zz / sigma[0]) ** 2 + (yy / sigma[1]) ** 2 + (xx / sigma[2]) ** 2))
    return (psf / psf.sum()).to(torch.float32).contiguous()


def make_tile(shape, bits_matrix, n_emitters, seed, device, kernels):
    """uint16 readouts + float32 predictor images of one synthetic tile, generated in HBM."""
    gen = torch.Generator(device=device).manual_seed(seed)
    C, B = bits_matrix.shape
    z = torch.randint(2, shape[0] - 2, (n_emitters,), device=device, generator=gen)
    y = torch.randint(4, shape[1] - 4, (n_emitters,), device=device, generator=gen)
    x = torch.randint(4, shape[2] - 4, (n_emitters,), device=device, generator=gen)
    code = torch.randint(0, C, (n_emitters,), device=device, generator=gen)
    amp = (torch.rand(n_emitters, device=device, generator=gen) * 600 + 300) / 0.0138  # 300-900 peak photons
    on = torch.from_numpy(bits_matrix).to(device)[code]  # [E, B]
    readouts, predictors = [], []
    scratch = torch.empty(shape, dtype=torch.float32, device=device)
    for b in range(B):
        sel = on[:, b] > 0
        vol = torch.zeros(shape, dtype=torch.float32, device=device)
        jitter = torch.rand(int(sel.sum()), device=device, generator=gen) * 0.4 + 0.8
        vol.index_put_((z[sel], y[sel], x[sel]), amp[sel] * jitter, accumulate=True)
        kernels.gaussian_lowpass(vol, (1.8, 1.6, 1.6), True, out=vol, scratch=scratch)  # PSF blur
        vol += 100.0
        img = torch.poisson(vol, generator=gen).clamp_(0, 65535).to(torch.uint16)
        readouts.append(img)
        pred = torch.zeros(shape, dtype=torch.float32, device=device)
        pred.index_put_((z[sel], y[sel], x[sel]), torch.ones(int(sel.sum()), device=device), accumulate=True)
        kernels.gaussian_lowpass(pred, (1.0, 1.2, 1.2), True, out=pred, scratch=scratch)
        pred.mul_(1.0 / 0.0441)  # blob peak -> 1
        pred += torch.rand(shape, device=device, generator=gen) * 0.02  # predictor floor (never exactly 0)
        pred.clamp_(0.0, 1.0)
        predictors.append(pred)
        del vol
    truth = torch.stack([code, z, y, x], dim=1).cpu().numpy()
    return readouts, predictors, truth


def cpu_baseline(readout0, psf, bits_matrix, tile_shape, n_bits, iters_per_bit):
    """Oracle (NumPy/SciPy/C restatement) timed on a bounded crop of the same workload."""
    from oracle import ccl_ref, decode_ref, rlgc_ref

    # sized for roughly 10-30 s of host work.  The transforms (most of the time) run on all the host cores
    # this process may use (SciPy pocketfft workers); NumPy's elementwise passes are single-threaded.
    cores = max(1, min(16, len(os.sched_getaffinity(0))))   # a one-GPU share of the host
    cz, cy, cx = min(tile_shape[0], 64), min(tile_shape[1], 256), min(tile_shape[2], 256)
    n_it, n_planes = 6, min(tile_shape[0], 16)
    crop = readout0[:cz, :cy, :cx].cpu().numpy()
    psf_np = psf.cpu().numpy()
    t0 = time.perf_counter()
    rlgc_ref.FFT_WORKERS = cores
    try:
        # rng="numpy": Generator.binomial in C, as a CPU implementation would draw the split (the oracle's
        # pure-NumPy Philox exists to check the HIP stream bit for bit and would dominate the time)
        _out, info = rlgc_ref.rlgc(crop, psf_np, rng="numpy", max_iterations=n_it, return_info=True)
    finally:
        rlgc_ref.FFT_WORKERS = 0
    t_rl = time.perf_counter() - t0
    np_crop = float(np.prod(info.padded_shape))
    # iterations + setup (~1 iteration's worth of FFTs): cost per padded voxel-iteration
    per_vox_iter = t_rl / ((info.iterations + 1.0) * np_crop)
    rl = importlib.import_module("merfish3d-analysis_amd.utils.rlgc")
    pads = rl.pad_widths_zyx(tile_shape, psf_np.shape)
    np_full = float(np.prod([s + a + b for s, (a, b) in zip(tile_shape, pads)]))
    # decode sample: n_planes x crop, all bits
    rng = np.random.default_rng(0)
    sample = rng.gamma(2.0, 50.0, size=(n_bits, n_planes, cy, cx)).astype(np.float32)
    cb = decode_ref.normalize_codebook(bits_matrix)
    t0 = time.perf_counter()
    lp = decode_ref.lowpass(sample, (3.0, 1.0, 1.0), True)
    dec, mag, dist, sc = decode_ref.decode_pixels(lp, cb, 0.7653669, (0.5, 10.0),
                                                  background=np.full(n_bits, 20, np.float32),
                                                  normalization=np.full(n_bits, 200, np.float32))
    ccl_ref.extract_barcodes(dec, sc, mag, dist, bits_matrix, [f"g{i}" for i in range(len(bits_matrix))],
                             0.6058, minimum_pixels=3)
    t_dec = time.perf_counter() - t0
    n_tile = float(np.prod(tile_shape))
    tile_seconds = n_bits * max(iters_per_bit, 1.0) * per_vox_iter * np_full + t_dec * n_tile / float(n_planes * cy * cx)
    return {
        "value": n_tile / tile_seconds / 1e6,
        "unit": "Mvoxels/s",
        "cores": cores,
        "kind": "port",
        "sample": (f"oracle RLGC (scipy.fft on {cores} threads, NumPy elementwise passes and Generator.binomial on one) {info.iterations} "
                   f"iterations on a {cz}x{cy}x{cx} crop (FFT volume "
                   f"{'x'.join(str(v) for v in info.padded_shape)}, {t_rl:.1f}s) + oracle low-pass/decode/extract on "
                   f"{n_planes}x{cy}x{cx}x{n_bits} bits ({t_dec:.1f}s), extrapolated linearly to {n_bits} bits x "
                   f"{iters_per_bit:.1f} iterations on the full tile"),
    }


def pmc_traffic(padded_shape, key="rlgc_iteration_hbm_bytes"):
    """HBM bytes per RLGC iteration (or per fft_conv launch) from the committed rocprofv3 PMC
    passes (FETCH_SIZE x 2 + WRITE_SIZE, profiles/r01_pmc_rlgc_traffic.json, made by
    scripts/pmc_summary.py) - only for the FFT volume they were collected on, else None.  PMC
    collection needs its own profiler passes, so this figure is measured offline, not in this run."""
    path = Path(__file__).resolve().parent / "profiles" / "r01_pmc_rlgc_traffic.json"
    try:
        d = json.loads(path.read_text())
    except (OSError, ValueError):
        return None
    if abs(d.get("padded_volume_bytes", 0) - 4.0 * float(np.prod(padded_shape))) > 1:
        return None
    return round(float(d[key]), 1) if d.get(key) else None


def aggregate_over_ranks(elapsed: float, iters: float, world: int, device) -> tuple[float, float]:
    """(max elapsed over ranks, total RLGC iterations over ranks) - the bench contract's timing rule."""
    if world <= 1:
        return float(elapsed), float(iters)
    import torch.distributed as dist

    if dist.get_backend() != "nccl":
        device = torch.device("cpu")
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    it = torch.tensor([iters], dtype=torch.float64, device=device)
    dist.all_reduce(it, op=dist.ReduceOp.SUM)
    return float(t.item()), float(it.item())


def io_leg(readouts, predictors, n_bits, step_seconds, voxels, sample_bits=4):
    """Datastore image I/O beside the compute: a sample of the tile's readout (uint16) and
    predictor (float32) images is written as OME-Zarr / Blosc-zstd stores and loaded back into HBM through
    ``zarr_io.ImageIO`` (zstd on host threads, bit shuffle + scatter on the GPU); per-tile figures are the
    sample scaled to all bits.  Reported next to ``value``, never part of it."""
    import shutil
    import tempfile

    zio = importlib.import_module("merfish3d-analysis_amd.zarr_io")
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    root = Path(tempfile.mkdtemp(prefix="m3d_bench_io_", dir=base))
    k = min(sample_bits, n_bits)
    try:
        with zio.ImageIO() as io:
            t0 = time.perf_counter()
            raw = files = 0
            for b in range(k):
                for name, img in (("corrected_data", readouts[b]), ("feature_predictor_data", predictors[b])):
                    io.write_image(root / f"bit{b + 1:03d}" / name, img)
                    raw += io.last_stats["raw_bytes"]
                    files += io.last_stats["file_bytes"]
            t_write = time.perf_counter() - t0
            t0 = time.perf_counter()
            ok = True
            for b in range(k):
                a = io.read_image(root / f"bit{b + 1:03d}" / "corrected_data", device=readouts[b].device)
                p = io.read_image(root / f"bit{b + 1:03d}" / "feature_predictor_data", device=readouts[b].device)
                ok = ok and a.shape == readouts[b].shape and p.dtype == predictors[b].dtype
            torch.cuda.synchronize()
            t_read = time.perf_counter() - t0
            ok = ok and torch.equal(a, readouts[k - 1]) and torch.equal(p, predictors[k - 1])
            threads = io.num_threads
    finally:
        shutil.rmtree(root, ignore_errors=True)
    load_tile = t_read * n_bits / k
    return {
        "what": f"{k} of {n_bits} bits: readout uint16 + predictor float32 images, OME-Zarr 0.5 / Zarr v3 / "
                f"Blosc zstd-5 bitshuffle chunks (16x512x512) on {'tmpfs' if base else 'local disk'}, device path",
        "host_threads": threads, "round_trip_equal": bool(ok),
        "read_GBps": round(raw / t_read / 1e9, 2), "write_GBps": round(raw / t_write / 1e9, 2),
        "compression_ratio": round(raw / max(files, 1), 2),
        "load_seconds_per_tile": round(load_tile, 3),
        "value_including_load": round(voxels / (step_seconds + load_tile) / 1e6, 4), "unit": "Mvoxels/s",
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="config2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # M3D_BENCH_BACKEND=gloo is a rehearsal switch (several ranks sharing one GPU on a one-GPU box,
    # which RCCL refuses); the measured configuration is always one rank per GPU over RCCL
    backend = os.environ.get("M3D_BENCH_BACKEND", "nccl")
    gpu_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(gpu_index)
    device = torch.device("cuda", gpu_index)
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    import __graft_entry__  # noqa: F401  (puts the repo root on sys.path)

    kernels = importlib.import_module("merfish3d-analysis_amd.kernels")
    rl = importlib.import_module("merfish3d-analysis_amd.utils.rlgc")
    PD = importlib.import_module("merfish3d-analysis_amd.PixelDecoder")
    DS = importlib.import_module("merfish3d-analysis_amd.datastore")

    shape, n_bits, psf_shape, n_emitters = WORKLOADS[args.workload]
    codebook = hw4_codebook(n_bits)
    bits_matrix = codebook.iloc[:, 1:].to_numpy(dtype=np.int64)
    psf = gaussian_psf(psf_shape, (1.8, 1.6, 1.6), device)
    readouts, predictors, truth = make_tile(shape, bits_matrix, n_emitters, 1 + rank, device, kernels)
    print(f"[bench] rank {rank}: synthetic tile ready", file=sys.stderr, flush=True)

    store = DS.InMemoryDataStore(num_tiles=1, num_rounds=1, num_bits=n_bits, codebook=codebook)
    for b in range(n_bits):
        store.set_corrected_image(readouts[b], tile=0, bit=b)
        store.save_local_feature_predictor_image(predictors[b], tile=0, bit=b)
    decoder = PD.PixelDecoder(store, merfish_bits=n_bits, verbose=0)
    decon = torch.empty(shape, dtype=torch.float32, device=device)
    stats = {"iters": 0, "conv_ms": 0.0, "conv_calls": 0, "transcripts": 0, "fused": True,
             "kernel_ms": [0.0] * 5, "loop_ms": 0.0}
    t_start = time.perf_counter()

    def progress(msg: str) -> None:  # keeps the GPU box's silence watchdog fed
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:6.1f}s] {msg}", file=sys.stderr, flush=True)

    def deconvolve_bits(profile: bool) -> None:
        for b in range(n_bits):  # DataRegistration._apply_bits_on_gpu: decon -> clip -> uint16
            _o, info = rl.rlgc_device(readouts[b], psf, rng_seed=42, out=decon, profile=profile)
            stats["iters"] += info.iterations
            stats["conv_ms"] += info.conv_ms_total
            stats["conv_calls"] += info.conv_calls
            stats["loop_ms"] += info.loop_ms
            stats["fused"] = info.fused_engine
            for k in range(5):
                stats["kernel_ms"][k] += info.kernel_ms[k]
            store.save_local_deconvolved_readout_image(kernels.clip_to_u16(decon), tile=0, bit=b)
            progress(f"bit {b + 1}/{n_bits}: {info.iterations} RLGC iterations ({info.stop_reason})")

    def step(profile: bool) -> None:
        deconvolve_bits(profile)
        decoder.decode_one_tile(tile_idx=0, gpu_id=gpu_index, normalization_method="global")
        stats["transcripts"] = len(decoder.decoded_barcodes)
        stats["table"] = decoder.decoded_barcodes[["tile_z", "tile_y", "tile_x", "barcode_id"]]
        progress(f"decode_one_tile: {stats['transcripts']} transcripts")

    def fix_normalization() -> None:
        """Per-dataset analysis parameters, set once before timing: background = median and
        normalisation = 99.9th percentile (minus background) of each bit's low-passed
        readout x predictor product (exact radix select on the device)."""
        decoder._tile_idx = 0
        decoder._load_bit_data(gpu_id=gpu_index)
        decoder._lp_filter(sigma=PD.DEFAULT_DECODE_LOWPASS_SIGMA, gpu_id=gpu_index)
        bkd = np.zeros(n_bits, dtype=np.float32)
        nrm = np.ones(n_bits, dtype=np.float32)
        for b in range(n_bits):
            pool = kernels.PooledValues(group=False)
            pool.add(decoder._image_data_lp[b])
            bkd[b] = pool.percentile(50.0)
            nrm[b] = max(pool.percentile(99.9) - bkd[b], 1e-3)
        store.save_decode_normalization_vectors(None, "global", nrm, bkd)
        decoder._cleanup()

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize(device)

    # untimed: first pass deconvolves so the per-dataset global normalisation vectors exist
    for w in range(max(args.warmup, 1)):
        if w == 0:
            deconvolve_bits(False)
            fix_normalization()
            decoder.decode_one_tile(tile_idx=0, gpu_id=gpu_index, normalization_method="global")
        else:
            step(False)
    stats.update({"iters": 0, "conv_ms": 0.0, "conv_calls": 0, "transcripts": 0, "kernel_ms": [0.0] * 5,
                  "loop_ms": 0.0})
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    sync()
    elapsed = time.perf_counter() - t0
    elapsed, total_iters = aggregate_over_ranks(elapsed, stats["iters"], world, device)

    if rank == 0:
        n_tile = float(np.prod(shape))
        value = world * args.steps * n_tile / elapsed / 1e6
        pads = rl.pad_widths_zyx(shape, psf_shape)
        padded = [s + a + b for s, (a, b) in zip(shape, pads)]
        np_full = float(np.prod(padded))
        conv_ms = stats["conv_ms"] / max(stats["conv_calls"], 1)
        conv_achieved = RLGC_ALGORITHMIC_BYTES_PER_CONV_VOXEL * np_full / (conv_ms * 1e-3) / 1e9 if conv_ms else 0.0
        # one RLGC iteration = the launch unit of the dominant work: SURVEY.md section 8(d) prices it at
        # 148 B per padded voxel (4 convolutions x 20 B + 17 elementwise array passes x 4 B)
        iter_ms = stats["loop_ms"] / max(float(stats["iters"]), 1.0)  # this rank's loops / this rank's iterations
        achieved = RLGC_ALGORITHMIC_BYTES_PER_ITER_VOXEL * np_full / (iter_ms * 1e-3) / 1e9 if iter_ms else 0.0
        out = {
            "metric": "Mvoxels/s through decon+decode, 22-bit codebook, 2048^2x64 tile",
            "value": round(value, 4),
            "unit": "Mvoxels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"configs[{1 if args.workload == 'config2' else 0}]: synthetic "
                             f"{shape[0]}x{shape[1]}x{shape[2]} tile per GPU, {n_bits}-bit/{len(bits_matrix)}-codeword "
                             f"codebook, {n_bits} x RLGC (PSF {'x'.join(map(str, psf_shape))}, FFT volume "
                             f"{'x'.join(map(str, padded))}) + decode_one_tile"),
                "tiles_per_step": world,
                "emitters_per_tile": n_emitters,
            },
            "rlgc_iterations_total": total_iters,
            "mvoxel_iterations_per_s": round(total_iters * n_tile / elapsed / 1e6, 2),
            "transcripts_rank0": stats["transcripts"],
            "roofline": {
                "kernel": ("one RLGC iteration = 4 x fft_conv (k_x_forward, k_y_pass, k_z_conv, k_y_pass, k_x_inverse; "
                           "the consensus and update arithmetic runs in two of the k_x_inverse store loops) "
                           "+ k_split_ratio + 2 one-block finalize kernels"),
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": pmc_traffic(padded),
                "avg_ms": round(iter_ms, 3),
                "launches": int(stats["iters"]),
                "algorithmic_bytes_per_launch": RLGC_ALGORITHMIC_BYTES_PER_ITER_VOXEL * np_full,
            },
            "roofline_conv": {
                "kernel": "fft_conv (5 kernels, 4 per iteration; two of the four carry fused elementwise tails)",
                "bound": "hbm", "achieved": round(conv_achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(conv_achieved / HBM_PEAK_GBS, 4),
                "traffic": pmc_traffic(padded, "fft_conv_hbm_bytes_per_launch"), "avg_ms": round(conv_ms, 3),
                "launches": stats["conv_calls"],
                "algorithmic_bytes_per_launch": RLGC_ALGORITHMIC_BYTES_PER_CONV_VOXEL * np_full,
            },
        }
        if stats["fused"] and stats["conv_calls"]:
            # the five kernels of one convolution, with the bytes each one actually has to move
            names = ("k_x_forward", "k_y_pass<fwd>", "k_z_conv", "k_y_pass<inv>", "k_x_inverse")
            # B per padded voxel each pass has to move (half spectrum = 4 B / voxel), averaged over the four
            # convolutions of an iteration (ConvBox in m3d_fftconv.h): estimate -> Hu (core rows in, mirrored in
            # LDS, core out), the two ratio convolutions (core in, full out, sharing one dual X inverse pass that
            # also runs the consensus product's X forward transform), consensus -> update (full in, core out)
            fz, fy, fx = (shape[i] / padded[i] for i in range(3))
            rows, vox = fz * fy, fz * fy * fx
            x_small = 4.0 * vox + 4.0 * rows   # real rows of the core <-> their half spectra
            y_small = 4.0 * rows + 4.0 * fz    # core z planes: core y rows <-> all ky
            x_dual = 8.0 + 4.0 + 4.0 + 4.0 * vox          # two spectra + update_norm in, product rows + HT sum out
            x_update = 4.0 * rows + 3.0 * 4.0 * vox        # consensus rows + estimate + HT sum in, estimate out
            moved = (3.0 * x_small / 4, (3.0 * y_small + 8.0) / 4, 5.0 * fz + 7.0, (y_small + 8.0) / 2,
                     (x_small + x_dual + x_update) / 4)
            out["roofline_kernels"] = [
                {"kernel": nme, "avg_ms": round(ms / stats["conv_calls"], 3),
                 "hbm_gbs": round(bpv * np_full / (ms / stats["conv_calls"] * 1e-3) / 1e9, 1) if ms else None,
                 "bytes_per_voxel": round(bpv, 2)}
                for nme, ms, bpv in zip(names, stats["kernel_ms"], moved)]
        # sanity of the decoded table against the planted emitters (voxel units, 3-voxel radius, same
        # codeword, one-to-one): the scorer of the simulation pipeline, not a timed quantity
        scoring = importlib.import_module("merfish3d-analysis_amd.scoring")
        table = stats.get("table")
        if table is not None:
            f1 = scoring.calculate_F1_with_radius(
                table[["tile_z", "tile_y", "tile_x"]].to_numpy(dtype=float), table["barcode_id"].to_numpy() - 1,
                truth[:, 1:4].astype(float), truth[:, 0], 3.0)[0]
            out["f1_vs_planted_emitters"] = {k: (round(v, 4) if isinstance(v, float) else int(v)) for k, v in f1.items()}
        # the one dense contraction of the path, outside the timed region: the fused decode kernel
        # (exact-f32 MFMA distance step) on this tile's low-passed bit stack, HIP events on the
        # stream it is launched on
        decoder._tile_idx = 0
        decoder._load_global_normalization_vectors(gpu_id=gpu_index)
        decoder._load_bit_data(gpu_id=gpu_index)
        decoder._lp_filter(sigma=PD.DEFAULT_DECODE_LOWPASS_SIGMA, gpu_id=gpu_index)
        decoder._decode_pixels(magnitude_threshold=PD.DEFAULT_DECODE_MAGNITUDE_THRESHOLD, gpu_id=gpu_index)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        reps = 3
        ev[0].record()
        for _ in range(reps):
            decoder._decode_pixels(magnitude_threshold=PD.DEFAULT_DECODE_MAGNITUDE_THRESHOLD, gpu_id=gpu_index)
        ev[1].record()
        torch.cuda.synchronize(device)
        dec_ms = ev[0].elapsed_time(ev[1]) / reps
        decoder._cleanup()
        flop = 2.0 * n_bits * len(bits_matrix) * n_tile
        out["roofline_decode"] = {
            "kernel": "k_decode (scale, clip, L2 norm, codeword distances on v_mfma_f32_32x32x2_f32, argmin, thresholds)",
            "bound": "mfma", "achieved": round(flop / (dec_ms * 1e-3) / 1e12, 2), "peak": F32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(flop / (dec_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
            "avg_ms": round(dec_ms, 3), "flop_per_launch": flop,
            "hbm_gbs": round((4.0 * n_bits + 6.0 + 2.0 * n_bits) * n_tile / (dec_ms * 1e-3) / 1e9, 1),
        }
        if world == 1 and not args.no_cpu_baseline:
            iters_per_bit = total_iters / max(args.steps * n_bits, 1)
            out["cpu_baseline"] = cpu_baseline(readouts[0], psf, bits_matrix, shape, n_bits, iters_per_bit)
            out["io"] = io_leg(readouts, predictors, n_bits, out["ms_per_step"] / 1e3, float(np.prod(shape)))
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
