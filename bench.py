#!/usr/bin/env python3
"""Headline benchmark: Mvoxels/s through decon + decode on synthetic MERFISH tiles.

One *step* = the per-tile hot path on one synthetic tile held in HBM.

``--workload config2`` (default, BASELINE.json ``configs[1]``): B x RLGC deconvolution of the readout
images (one per bit, clip -> uint16 like DataRegistration.py:993-995) followed by one full
``PixelDecoder.decode_one_tile`` (readout x predictor, low-pass, fused pixel decode, connected components,
transcript table) on a 64 x 2048 x 2048 tile, 22-bit / 140-codeword codebook, PSF 64 x 51 x 51 (FFT volume
128 x 2187 x 2187).  Every rank works on its own tile per step: weak scaling, no data-path collective.

``--workload config3`` (``configs[2]``): a volume of ``--tiles`` such tiles with 11 rounds each, dealt to the
ranks in contiguous ranges (PixelDecoder.py:4811-4820, DataRegistration.py:2139-2156): per tile 11 fiducial
deconvolutions, 10 x ``register_pair_to_fixed`` against round 1, 22 readout deconvolutions and
``decode_one_tile`` warping every bit with its round's transform.  The tile count is fixed as N grows:
strong scaling.  Tiles are synthesised outside the timed regions (one region per tile, device-synchronised
on both sides), so the inputs are resident in HBM whenever the clock runs.

``--store disk`` (round 4; SURVEY.md 8(e)'s strong-scaling risk measured on one GPU): the tiles live in an on-disk qi2lab
datastore (tmpfs if the box has one, a directory per rank).  Per tile the timed region then reads the fiducial, readout
and predictor images (read-ahead of tile t + 1 under tile t's kernels) and writes the deconvolved volumes and the
transcript table (write-behind), on ``--io-read-threads`` + ``--io-write-threads`` zstd threads.  In this mode ``value`` IS
the wall-clock rate including every wait for I/O and the final drain of the write queue; the JSON's ``store`` object
repeats it as ``value_including_io`` next to ``value_compute_only`` (the main thread's I/O waits subtracted),
``caller_wait_s_per_tile``, ``background_s_per_tile``, ``host_cpu_s_per_tile`` (process CPU time), ``io_threads`` and
``round_trip_equal`` (a volume read back from disk equals what was computed).  The HBM-resident figure to hold it
against is the same command without ``--store disk`` (61.6 s against 61.2 s per tile).

``card_state_rank0`` (diagnostic): shader / memory clock, board power and temperatures of rank 0's card sampled from its
hwmon files every two seconds while the steps run; with more than one rank ``card_state_all_ranks`` adds the lowest and
highest mean clock, power and HBM temperature over the ranks' cards (the max-over-ranks rule makes the slowest card's
clock the job's).  The workload holds the card at its power limit and the shader clock
is what the firmware gives way with - it explains the +-1.5 % between runs of the same tree.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8                      # starts its own 8 ranks (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

**Wall budget.**  A config-2 tile takes tens of seconds, so ``--steps`` / ``--warmup`` are upper bounds under a
wall-clock budget counted from process start (``M3D_BENCH_WALL_S``, default 450 s): the first (untimed) tile is
timed, the number of timed tiles that fit is derived from it, and the JSON reports ``steps`` = tiles actually
timed next to ``steps_requested``.  The CPU baseline, the decode-kernel timing and the I/O leg run *before* the
timed loop, and rank 0 prints a complete JSON line after every timed tile (the last line wins), so a killed run
still leaves a parsed record.

``roofline`` prices one RLGC iteration (the unit that is launched ~1600 times per tile) at SURVEY.md section
8(d)'s 148 algorithmic bytes per padded voxel over the iteration's device time measured with HIP events inside
``m3d_rlgc_run``; ``roofline_conv`` and ``roofline_kernels`` break it down, ``roofline_decode`` is the MFMA-bound
decode kernel.
"""

from __future__ import annotations

import time

_T_IMPORT0 = time.time()   # the wall budget counts from here (or from the launching parent's start)

import argparse  # noqa: E402
import importlib  # noqa: E402
import itertools  # noqa: E402
import json  # noqa: E402
import os  # noqa: E402
import socket  # noqa: E402
import subprocess  # noqa: E402
import sys  # noqa: E402
from pathlib import Path  # noqa: E402

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import torch  # noqa: E402

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

WORKLOADS = {
    # name: (tile zyx, bits, psf zyx, emitters, rounds)
    "config2": ((64, 2048, 2048), 22, (64, 51, 51), 80_000, 1),
    "config1": ((16, 256, 256), 16, (16, 51, 51), 1_200, 1),
    "config3": ((64, 2048, 2048), 22, (64, 51, 51), 80_000, 11),
    "config3_small": ((16, 256, 256), 16, (16, 51, 51), 1_200, 8),   # rehearsal size of config3
}
CONFIG_INDEX = {"config1": 0, "config2": 1, "config3": 2, "config3_small": 2}
RLGC_ALGORITHMIC_BYTES_PER_ITER_VOXEL = 148.0
RLGC_ALGORITHMIC_BYTES_PER_CONV_VOXEL = 20.0  # 2 FFTs x 8 B + OTF read 4 B
F32_MFMA_PEAK_TFLOPS = 157.3  # dense fp32 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
DEFAULT_WALL_BUDGET_S = 450.0
MISREGISTRATION_PX = 0.35   # the reference's own acceptance bound (tests/test_sofima_deformable_registration.py:175-218)


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


def make_tile(shape, bits_matrix, n_emitters, seed, device, kernels, round_shifts_px=None, n_beads=0):
    """uint16 readouts + float32 predictor images of one synthetic tile, generated in HBM.

    ``round_shifts_px`` (config 3): one (dz, dy, dx) per round; bit b belongs to round ``b * R // B`` and its
    readout / predictor images are the round-1-frame images displaced by that round's shift (trilinear
    resample), and a fiducial bead volume per round is returned as well."""
    gen = torch.Generator(device=device).manual_seed(seed)
    C, B = bits_matrix.shape
    z = torch.randint(2, shape[0] - 2, (n_emitters,), device=device, generator=gen)
    y = torch.randint(4, shape[1] - 4, (n_emitters,), device=device, generator=gen)
    x = torch.randint(4, shape[2] - 4, (n_emitters,), device=device, generator=gen)
    code = torch.randint(0, C, (n_emitters,), device=device, generator=gen)
    amp = (torch.rand(n_emitters, device=device, generator=gen) * 600 + 300) / 0.0138  # 300-900 peak photons
    on = torch.from_numpy(bits_matrix).to(device)[code]  # [E, B]
    readouts, predictors, fiducials = [], [], []
    scratch = torch.empty(shape, dtype=torch.float32, device=device)
    n_rounds = len(round_shifts_px) if round_shifts_px is not None else 1

    def displaced(vol, r):
        # content of the round-1 frame seen in round r: sample at x - shift (pull offset = -shift)
        if round_shifts_px is None or not any(round_shifts_px[r]):
            return vol
        off = [-float(v) for v in round_shifts_px[r]]
        return kernels.affine_warp(vol, np.eye(3), off, shape, mode="nearest")

    for b in range(B):
        r = b * n_rounds // B
        sel = on[:, b] > 0
        vol = torch.zeros(shape, dtype=torch.float32, device=device)
        jitter = torch.rand(int(sel.sum()), device=device, generator=gen) * 0.4 + 0.8
        vol.index_put_((z[sel], y[sel], x[sel]), amp[sel] * jitter, accumulate=True)
        kernels.gaussian_lowpass(vol, (1.8, 1.6, 1.6), True, out=vol, scratch=scratch)  # PSF blur
        vol = displaced(vol, r)
        vol += 100.0
        img = torch.poisson(vol, generator=gen).clamp_(0, 65535).to(torch.uint16)
        readouts.append(img)
        pred = torch.zeros(shape, dtype=torch.float32, device=device)
        pred.index_put_((z[sel], y[sel], x[sel]), torch.ones(int(sel.sum()), device=device), accumulate=True)
        kernels.gaussian_lowpass(pred, (1.0, 1.2, 1.2), True, out=pred, scratch=scratch)
        pred = displaced(pred, r)
        pred.mul_(1.0 / 0.0441)  # blob peak -> 1
        pred += torch.rand(shape, device=device, generator=gen) * 0.02  # predictor floor (never exactly 0)
        pred.clamp_(0.0, 1.0)
        predictors.append(pred)
        del vol
    if round_shifts_px is not None:
        nb = int(n_beads)
        bz = torch.randint(2, shape[0] - 2, (nb,), device=device, generator=gen)
        by = torch.randint(8, shape[1] - 8, (nb,), device=device, generator=gen)
        bx = torch.randint(8, shape[2] - 8, (nb,), device=device, generator=gen)
        bamp = (torch.rand(nb, device=device, generator=gen) * 2000 + 2000) / 0.0138
        base = torch.zeros(shape, dtype=torch.float32, device=device)
        base.index_put_((bz, by, bx), bamp, accumulate=True)
        kernels.gaussian_lowpass(base, (1.8, 1.6, 1.6), True, out=base, scratch=scratch)
        for r in range(n_rounds):
            vol = displaced(base, r).clone()
            vol += 100.0
            fiducials.append(torch.poisson(vol, generator=gen).clamp_(0, 65535).to(torch.uint16))
            del vol
    truth = torch.stack([code, z, y, x], dim=1).cpu().numpy()
    return readouts, predictors, truth, fiducials


def cpu_baseline(readout0, psf, bits_matrix, tile_shape, n_bits, iters_per_bit, rlgc_calls_per_tile=None):
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
    calls = n_bits if rlgc_calls_per_tile is None else rlgc_calls_per_tile
    tile_seconds = calls * max(iters_per_bit, 1.0) * per_vox_iter * np_full + t_dec * n_tile / float(n_planes * cy * cx)
    return {
        "value": n_tile / tile_seconds / 1e6,
        "unit": "Mvoxels/s",
        "cores": cores,
        "kind": "port",
        "sample": (f"oracle RLGC (scipy.fft on {cores} threads, NumPy elementwise passes and Generator.binomial on one) {info.iterations} "
                   f"iterations on a {cz}x{cy}x{cx} crop (FFT volume "
                   f"{'x'.join(str(v) for v in info.padded_shape)}, {t_rl:.1f}s) + oracle low-pass/decode/extract on "
                   f"{n_planes}x{cy}x{cx}x{n_bits} bits ({t_dec:.1f}s), extrapolated linearly to {calls} deconvolutions x "
                   f"{iters_per_bit:.1f} iterations on the full tile"),
    }


def pmc_traffic(padded_shape, key="rlgc_iteration_hbm_bytes"):
    """HBM bytes per RLGC iteration (or per fft_conv launch) from the committed rocprofv3 PMC
    passes (FETCH_SIZE x 2 + WRITE_SIZE, profiles/rNN_pmc_rlgc_traffic.json, made by
    scripts/pmc_summary.py) - only for the FFT volume they were collected on, else None.  PMC
    collection needs its own profiler passes, so this figure is measured offline, not in this run."""
    for path in sorted((Path(__file__).resolve().parent / "profiles").glob("r*_pmc_rlgc_traffic.json"),
                       reverse=True):
        try:
            d = json.loads(path.read_text())
        except (OSError, ValueError):
            continue
        if abs(d.get("padded_volume_bytes", 0) - 4.0 * float(np.prod(padded_shape))) > 1:
            continue
        if d.get(key):
            return round(float(d[key]), 1)
    return None


def _process_group_live() -> bool:
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized()


def aggregate_over_ranks(elapsed: float, iters: float, world: int, device) -> tuple[float, float]:
    """(max elapsed over ranks, total RLGC iterations over ranks) - the bench contract's timing rule.  A live
    process group is used whatever its size (a one-rank RCCL group is how a one-GPU box rehearses this)."""
    if world <= 1 and not _process_group_live():
        return float(elapsed), float(iters)
    import torch.distributed as dist

    if dist.get_backend() != "nccl":
        device = torch.device("cpu")
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    it = torch.tensor([iters], dtype=torch.float64, device=device)
    dist.all_reduce(it, op=dist.ReduceOp.SUM)
    return float(t.item()), float(it.item())


def card_state_over_ranks(summary, world: int, device):
    """The ranks' card states side by side: lowest and highest mean shader clock, board power and HBM temperature over
    the ranks (a rank without a record counts for nothing) - the max-over-ranks timing rule makes the slowest card's
    clock the job's.  Collective: every rank calls it."""
    keys = ("sclk_mhz", "power_w", "hbm_c")
    mine = [float(summary[k]["mean"]) if summary and k in summary else float("nan") for k in keys]
    if world <= 1 and not _process_group_live():
        lo, hi, n = mine, mine, 1 if summary else 0
    else:
        import torch.distributed as dist

        if dist.get_backend() != "nccl":
            device = torch.device("cpu")
        have = 0.0 if any(v != v for v in mine) else 1.0
        big = 1e30
        t_lo = torch.tensor([v if have else big for v in mine], dtype=torch.float64, device=device)
        t_hi = torch.tensor([v if have else -big for v in mine], dtype=torch.float64, device=device)
        t_n = torch.tensor([have], dtype=torch.float64, device=device)
        dist.all_reduce(t_lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(t_hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(t_n, op=dist.ReduceOp.SUM)
        lo, hi, n = t_lo.tolist(), t_hi.tolist(), int(t_n.item())
    if not n:
        return None
    return {"ranks_reporting": n, **{k: {"lowest_rank_mean": round(a, 1), "highest_rank_mean": round(b, 1)}
                                       for k, a, b in zip(keys, lo, hi)}}


def agree_min(value: int, world: int, device) -> int:
    """The same integer on every rank: the minimum of the ranks' proposals (step counts must agree)."""
    if world <= 1 and not _process_group_live():
        return int(value)
    import torch.distributed as dist

    if dist.get_backend() != "nccl":
        device = torch.device("cpu")
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def steps_that_fit(requested: int, seconds_per_step: float, seconds_left: float, margin: float = 1.08) -> int:
    """How many of ``requested`` steps fit into ``seconds_left`` (at least one: a bench that times nothing
    reports nothing)."""
    if seconds_per_step <= 0:
        return max(1, int(requested))
    return max(1, min(int(requested), int(seconds_left / (seconds_per_step * margin))))


def contiguous_share(n_items: int, world: int, rank: int) -> range:
    """Contiguous range of ``n_items`` for ``rank`` (the reference's chunking, PixelDecoder.py:4811-4820)."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def io_leg(readouts, predictors, n_bits, sample_bits=4):
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
        "load_seconds_per_tile": round(load_tile, 3), "unit": "Mvoxels/s",
    }


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


class CardState:
    """Clocks, board power and temperatures of one card while the timed region runs - a diagnostic next to ``value``,
    not part of it.  The card runs this workload AT ITS POWER LIMIT (about 1 350 W: the shader clock sits near 2.1 of
    2.4 GHz), and a single run carries +-1.5 % from one process to the next (``profiles/r04_workspace_placement.txt``);
    these numbers say which state a run met.  Plain reads of the amdgpu hwmon files of the card's PCI function every
    two seconds from a host thread (``rocm-smi`` reads the same files); any failure leaves the record out."""

    FILES = {"sclk_mhz": ("freq1_input", 1e-6), "mclk_mhz": ("freq2_input", 1e-6), "power_w": ("power1_input", 1e-6),
             "junction_c": ("temp2_input", 1e-3), "hbm_c": ("temp3_input", 1e-3)}

    def __init__(self, hwmon_dir, period_s: float = 2.0):
        import threading

        self.dir = Path(hwmon_dir) if hwmon_dir else None
        self.period = period_s
        self.samples: dict = {k: [] for k in self.FILES}
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True) if self.dir else None

    @staticmethod
    def hwmon_of(device_index: int, sysfs: str = "/sys/bus/pci/devices"):
        try:
            pr = torch.cuda.get_device_properties(device_index)
            fn = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            found = sorted(Path(sysfs, fn, "hwmon").glob("hwmon*"))
            return found[0] if found else None
        except Exception:
            return None

    def _read(self, name: str, scale: float):
        try:
            return float((self.dir / name).read_text()) * scale
        except (OSError, ValueError):
            return None

    def _run(self) -> None:
        while not self._stop.wait(self.period):
            for key, (name, scale) in self.FILES.items():
                v = self._read(name, scale)
                if v is not None:
                    self.samples[key].append(v)

    def start(self) -> "CardState":
        if self._thread:
            self._thread.start()
        return self

    def stop(self) -> None:
        if self._thread and self._thread.is_alive():
            self._stop.set()
            self._thread.join(timeout=5.0)

    def summary(self):
        """-> {"sclk_mhz": {"mean", "min", "max"}, ..., "power_cap_w", "samples"} of the samples so far, or None."""
        if not self._thread:
            return None
        out = {k: {"mean": round(float(np.mean(v)), 1), "min": round(min(v), 1), "max": round(max(v), 1)}
               for k, v in ((k, list(v)) for k, v in self.samples.items()) if v}
        if not out:
            return None
        cap = self._read("power1_cap", 1e-6)
        if cap is not None:
            out["power_cap_w"] = round(cap, 1)
        out["samples"] = max(len(v) for v in self.samples.values())
        return out


def visible_gpu_count() -> int | None:
    """GPUs this process may use, found WITHOUT a HIP call (the launching parent must stay clear of the runtime): the
    smallest of the visibility lists that are set, capped by the compute nodes of the KFD topology (nodes with SIMDs
    are GPUs) where that can be read.  ``None`` = cannot tell (no list, no readable topology - some containers): the
    caller then skips its pre-check and lets the ranks fail with their own error."""
    listed = []
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            listed.append(len([x for x in v.split(",") if x.strip() != ""]))
    nodes = None
    try:
        props_files = list(Path("/sys/class/kfd/kfd/topology/nodes").glob("*/properties"))
    except OSError:
        props_files = []
    if props_files:
        nodes = 0
        for props in props_files:
            try:
                fields = dict(line.split()[:2] for line in props.read_text().splitlines() if len(line.split()) >= 2)
                nodes += int(fields.get("simd_count", "0")) > 0
            except (OSError, ValueError):
                nodes = None   # unreadable topology: no claim
                break
    if listed:
        return min(listed) if nodes is None else min(min(listed), nodes)
    return nodes


def launch_ranks(n: int, argv: list[str]) -> int:
    """``bench.py --gpus N`` started plainly: start N ranks with torch.distributed.run as a CHILD process and
    pass its output and exit code on.  The parent has made no GPU call (GPUs are counted from the environment /
    sysfs) and never execs."""
    have = visible_gpu_count()
    if have is not None and have < n and os.environ.get("M3D_BENCH_BACKEND", "nccl") == "nccl":   # (gloo rehearsal: one GPU)
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible on this node")
    env = dict(os.environ)
    env["M3D_BENCH_T0"] = repr(_T_IMPORT0)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(Path(__file__).resolve()), *argv]
    print(f"[bench] WORLD_SIZE unset and --gpus {n}: starting {n} ranks: {' '.join(cmd[1:8])} ...",
          file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="config2")
    ap.add_argument("--tiles", type=int, default=8, help="config3: tiles of the volume (fixed as N grows)")
    ap.add_argument("--sofima", action="store_true",
                    help="config3: also run the SOFIMA residual-flow estimator per moving round (configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--store", choices=("hbm", "disk"), default="hbm",
                    help="disk: every tile comes from and goes back to an on-disk qi2lab datastore (tmpfs if there is "
                         "one): tile t+1's images are read ahead and tile t's deconvolved volumes and transcript table "
                         "written behind, on host threads and side streams under tile t's kernels")
    ap.add_argument("--io-read-threads", type=int, default=4, help="--store disk: zstd threads of the read-ahead")
    ap.add_argument("--io-write-threads", type=int, default=10, help="--store disk: zstd threads of the write-behind")
    ap.add_argument("--io-queue", type=int, default=6, help="--store disk: volumes the write-behind queue may hold in HBM")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus, sys.argv[1:])

    t_origin = float(os.environ.get("M3D_BENCH_T0", _T_IMPORT0))
    budget = float(os.environ.get("M3D_BENCH_WALL_S", DEFAULT_WALL_BUDGET_S))

    def seconds_left() -> float:
        return budget - (time.time() - t_origin)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: every requested GPU needs its own rank")
    # M3D_BENCH_BACKEND=gloo is a rehearsal switch (several ranks sharing one GPU on a one-GPU box,
    # which RCCL refuses); the measured configuration is always one rank per GPU over RCCL
    backend = os.environ.get("M3D_BENCH_BACKEND", "nccl")
    gpu_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(gpu_index)
    device = torch.device("cuda", gpu_index)
    # M3D_BENCH_FORCE_DIST=1: a one-rank run still joins an RCCL process group, so the barrier and the
    # all-reduces of the N > 1 path execute on a one-GPU box (tests/test_gpu_rccl_single_rank.py)
    force_dist = world == 1 and os.environ.get("M3D_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist

        if force_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"only {dist.get_world_size()} of {args.gpus} ranks joined")

    import __graft_entry__  # noqa: F401  (puts the repo root on sys.path)

    kernels = importlib.import_module("merfish3d-analysis_amd.kernels")
    rl = importlib.import_module("merfish3d-analysis_amd.utils.rlgc")
    mv = importlib.import_module("merfish3d-analysis_amd.utils.multiview_registration")
    sofima = importlib.import_module("merfish3d-analysis_amd.utils.sofima_registration")
    PD = importlib.import_module("merfish3d-analysis_amd.PixelDecoder")
    DS = importlib.import_module("merfish3d-analysis_amd.datastore")
    scoring = importlib.import_module("merfish3d-analysis_amd.scoring")

    shape, n_bits, psf_shape, n_emitters, n_rounds = WORKLOADS[args.workload]
    strong = n_rounds > 1   # config 3: a fixed volume of tiles dealt to the ranks
    codebook = hw4_codebook(n_bits)
    bits_matrix = codebook.iloc[:, 1:].to_numpy(dtype=np.int64)
    psf = gaussian_psf(psf_shape, (1.8, 1.6, 1.6), device)
    n_tile = float(np.prod(shape))
    pads = rl.pad_widths_zyx(shape, psf_shape)
    padded = [s + a + b for s, (a, b) in zip(shape, pads)]
    np_full = float(np.prod(padded))
    spacing = (0.315, 0.098, 0.098)
    rlgc_calls_per_tile = n_bits + (n_rounds if strong else 0)

    disk = args.store == "disk"
    io_root = None
    if disk:
        # SURVEY.md 8(e)'s strong-scaling risk, measured where it can be: the tile's images really come from and go
        # back to a datastore directory (qi2labDataStore.py:2234-2373 layout and codecs), overlapped with the compute
        import tempfile

        QS = importlib.import_module("merfish3d-analysis_amd.qi2lab_store")
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        io_root = Path(tempfile.mkdtemp(prefix=f"m3d_bench_store_r{rank}_", dir=base))
        n_slots = len(contiguous_share(args.tiles, world, rank)) if strong else 2   # weak: the rank's tile twice, see slot_of
        store = QS.Qi2labDiskStore(io_root, device=device, num_threads=4, write_behind=True,
                                   prefetch_threads=args.io_read_threads, write_threads=args.io_write_threads,
                                   max_pending_writes=args.io_queue,
                                   num_tiles=max(n_slots, 1), num_rounds=n_rounds, num_bits=n_bits, codebook=codebook,
                                   voxel_size_zyx_um=spacing)
    else:
        store = DS.InMemoryDataStore(num_tiles=1, num_rounds=n_rounds, num_bits=n_bits, codebook=codebook,
                                     voxel_size_zyx_um=spacing)
    decoder = PD.PixelDecoder(store, merfish_bits=n_bits, verbose=0)
    decon = torch.empty(shape, dtype=torch.float32, device=device)
    stats: dict = {}

    def reset_stats() -> None:
        stats.update({"iters": 0, "conv_ms": 0.0, "conv_calls": 0, "transcripts": 0, "fused": True,
                      "kernel_ms": [0.0] * 5, "loop_ms": 0.0, "reg_s": 0.0, "reg_calls": 0, "reg_err_px": 0.0,
                      "reg_max_s": 0.0, "reg_off": 0, "truth": stats.get("truth"), "phase_s": {}})

    reset_stats()
    t_start = time.perf_counter()

    def progress(msg: str) -> None:  # keeps the GPU box's silence watchdog fed
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:6.1f}s] {msg}", file=sys.stderr, flush=True)

    def tile_shifts(tile_seed: int):
        """config 3: per-round shifts <= (1, 4, 4) px, round 1 at rest."""
        rng = np.random.default_rng(tile_seed)
        s = rng.uniform(-1.0, 1.0, size=(n_rounds, 3)) * np.asarray([1.0, 4.0, 4.0])
        s[0] = 0.0
        return np.round(s, 2)

    current = {}
    slot_truth: dict = {}   # --store disk: per store tile, what the generator planted (truth table, round shifts)

    def load_tile(tile_seed: int, slot: int = 0) -> None:
        """Synthesise one tile in HBM and put it into the store (outside every timed region).  ``--store disk``: the
        tile's input images are written to the datastore directory as tile ``slot`` and dropped from HBM."""
        current.clear()
        if not disk:
            store._readout_decon.clear()
            store._fiducial_decon.clear()
            store._round_xform.clear()
        shifts = tile_shifts(tile_seed) if strong else None
        readouts, predictors, truth, fiducials = make_tile(
            shape, bits_matrix, n_emitters, tile_seed, device, kernels, round_shifts_px=shifts,
            n_beads=max(300, int(n_tile / 3500)))   # config-1 density: 300 beads in 16x256x256
        for b in range(n_bits):
            store.set_corrected_image(readouts[b], tile=slot, bit=b)
            store.save_local_feature_predictor_image(predictors[b], tile=slot, bit=b)
            store.set_round_linker(store.tile_ids[slot], store.bit_ids[b], 1 + b * n_rounds // n_bits)
        for r, f in enumerate(fiducials):
            store.set_corrected_image(f, tile=slot, round=store.round_ids[r])
        if disk:
            store.flush()
            store.evict_tile(slot)          # inputs are on disk only: the timed region has to bring them in
            slot_truth[slot] = dict(truth=truth, shifts=shifts)
            current.update(truth=truth, shifts=shifts)
            del readouts, predictors, fiducials
        else:
            current.update(readouts=readouts, predictors=predictors, truth=truth, fiducials=fiducials, shifts=shifts)
        torch.cuda.synchronize(device)

    def fiducial_image(slot: int, r: int):
        if disk:
            return store.load_local_corrected_image(tile=slot, round=store.round_ids[r], return_future=False)
        return current["fiducials"][r]

    def readout_image(slot: int, b: int):
        if disk:
            return store.load_local_corrected_image(tile=slot, bit=store.bit_ids[b], return_future=False)
        return current["readouts"][b]

    def deconvolve(image, profile):
        # profile: True = HIP events around every FFT pass (the warm-up tile: per-pass breakdown), "loop" = only the
        # two events around each solver call's iteration loop (the timed tiles: roofline.avg_ms)
        _o, info = rl.rlgc_device(image, psf, rng_seed=42, out=decon, profile=profile)
        stats["iters"] += info.iterations
        stats["conv_ms"] += info.conv_ms_total
        stats["conv_calls"] += info.conv_calls
        stats["loop_ms"] += info.loop_ms
        stats["fused"] = info.fused_engine
        for k in range(5):
            stats["kernel_ms"][k] += info.kernel_ms[k] if info.kernel_ms else 0.0
        return kernels.clip_to_u16(decon), info

    def register_rounds(profile: bool, slot: int = 0) -> None:
        """DataRegistration._process_fiducial_rounds_on_gpu (:456-558) with the volumes staying in HBM."""
        t = store.tile_ids[slot]
        shifts = slot_truth[slot]["shifts"] if disk else current["shifts"]
        reference, _ = deconvolve(fiducial_image(slot, 0), profile)
        if disk:
            store.save_local_deconvolved_fiducial_image(reference, tile=t, round=store.round_ids[0])
        store.save_local_round_transform_zyx_um(np.eye(4, dtype=np.float32), tile=t, round=store.round_ids[0])
        for r in range(1, n_rounds):
            moving, info = deconvolve(fiducial_image(slot, r), profile)
            if disk:
                store.save_local_deconvolved_fiducial_image(moving, tile=t, round=store.round_ids[r])
            t0 = time.perf_counter()
            T = mv.register_pair_to_fixed(reference, moving, spacing_zyx_um=spacing)
            torch.cuda.synchronize(device)
            dt = time.perf_counter() - t0
            stats["reg_s"] += dt
            stats["reg_calls"] += 1
            stats["reg_max_s"] = max(stats.get("reg_max_s", 0.0), dt)
            store.save_local_round_transform_zyx_um(np.asarray(T, dtype=np.float32), tile=t, round=store.round_ids[r])
            # moving = reference displaced by +shift => pull transform = -shift ... in the generator's
            # convention the content sits at x + shift, so the pull offset is +shift
            err = np.abs(T[:3, 3] / np.asarray(spacing) - shifts[r]).max()
            stats["reg_err_px"] = max(stats["reg_err_px"], float(err))
            stats["reg_off"] += int(err > MISREGISTRATION_PX)
            progress(f"round {r + 1}/{n_rounds}: {info.iterations} RLGC iterations, shift error {err:.3f} px")
            if args.sofima:
                # DataRegistration._process_sofima_rounds_on_gpu (:561-693): residual flow after the affine
                t0 = time.perf_counter()
                initialized = mv.warp_array_to_reference_gpu(moving, transform_zyx_um=np.asarray(T, dtype=np.float32),
                                                             spacing_zyx_um=spacing, reference_shape=shape)
                _flow, md = sofima.estimate_sofima_flow_field_xyz_px(reference, initialized)
                torch.cuda.synchronize(device)
                stats["sofima_s"] = stats.get("sofima_s", 0.0) + time.perf_counter() - t0
                stats["sofima_calls"] = stats.get("sofima_calls", 0) + 1
                stats["sofima_status"] = md["status"]
                del initialized, _flow

    def deconvolve_bits(profile: bool, slot: int = 0) -> None:
        for b in range(n_bits):  # DataRegistration._apply_bits_on_gpu: decon -> clip -> uint16
            u16, info = deconvolve(readout_image(slot, b), profile)
            store.save_local_deconvolved_readout_image(u16, tile=slot, bit=b)
            if disk and b == 0:
                stats["kept_u16"] = (slot, u16)      # for the byte-for-byte read-back check after the run
            progress(f"bit {b + 1}/{n_bits}: {info.iterations} RLGC iterations ({info.stop_reason})")

    def decode(slot: int = 0) -> None:
        t0 = time.perf_counter()
        decoder.decode_one_tile(tile_idx=slot, gpu_id=gpu_index, normalization_method="global")
        torch.cuda.current_stream(device).synchronize()
        stats["phase_s"]["decode_one_tile"] = stats["phase_s"].get("decode_one_tile", 0.0) + time.perf_counter() - t0
        stats["transcripts"] = len(decoder.decoded_barcodes)
        stats["table"] = decoder.decoded_barcodes[["tile_z", "tile_y", "tile_x", "barcode_id"]]
        stats["truth"] = slot_truth[slot]["truth"] if disk else current["truth"]
        if disk:
            store.save_local_decoded_spots(decoder.decoded_barcodes, tile=slot)   # PixelDecoder.py:3203-3233
        progress(f"decode_one_tile: {stats['transcripts']} transcripts")

    def step(profile: bool, slot: int = 0, next_slot: int | None = None) -> None:
        if disk and next_slot is not None:
            store.prefetch_tile_inputs(next_slot)     # read-ahead of the next tile under this tile's kernels
        t0 = time.perf_counter()
        if strong:
            register_rounds(profile, slot)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        deconvolve_bits(profile, slot)
        torch.cuda.synchronize(device)
        t2 = time.perf_counter()
        decode(slot)
        if disk:
            store.evict_tile(slot)                    # queued writes keep their volumes until they are on disk
        torch.cuda.synchronize(device)
        t3 = time.perf_counter()
        for name, dt in (("fiducials_and_registration", t1 - t0), ("readout_deconvolution", t2 - t1), ("decode", t3 - t2)):
            stats["phase_s"][name] = stats["phase_s"].get(name, 0.0) + dt

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
        if world > 1 or force_dist:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize(device)

    def decode_kernel_leg() -> dict:
        """The one dense contraction of the path, outside the timed region: the fused decode kernel (exact-f32
        MFMA distance step) on this tile's low-passed bit stack, HIP events on the stream it is launched on."""
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
        return {
            "kernel": "k_decode (scale, clip, L2 norm, codeword distances on v_mfma_f32_32x32x2_f32, argmin, thresholds)",
            "bound": "mfma", "achieved": round(flop / (dec_ms * 1e-3) / 1e12, 2), "peak": F32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(flop / (dec_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
            "avg_ms": round(dec_ms, 3), "flop_per_launch": flop,
            "hbm_gbs": round((4.0 * n_bits + 6.0 + 2.0 * n_bits) * n_tile / (dec_ms * 1e-3) / 1e9, 1),
        }

    extras: dict = {}

    def record(steps_done: int, warm_done: int, elapsed: float, total_iters: float, tiles_done: int) -> dict:
        value = tiles_done * n_tile / elapsed / 1e6
        conv_ms = passes["conv_ms"] / max(passes["conv_calls"], 1)
        conv_achieved = RLGC_ALGORITHMIC_BYTES_PER_CONV_VOXEL * np_full / (conv_ms * 1e-3) / 1e9 if conv_ms else 0.0
        # one RLGC iteration = the launch unit of the dominant work: SURVEY.md section 8(d) prices it at
        # 148 B per padded voxel (4 convolutions x 20 B + 17 elementwise array passes x 4 B)
        iter_ms = stats["loop_ms"] / max(float(stats["iters"]), 1.0)  # this rank's loops / this rank's iterations
        achieved = RLGC_ALGORITHMIC_BYTES_PER_ITER_VOXEL * np_full / (iter_ms * 1e-3) / 1e9 if iter_ms else 0.0
        what = (f"{n_bits} x RLGC (PSF {'x'.join(map(str, psf_shape))}, FFT volume {'x'.join(map(str, padded))}) + "
                f"decode_one_tile")
        if strong:
            what = (f"{n_rounds} fiducial RLGC + {n_rounds - 1} x register_pair_to_fixed + " + what +
                    " with per-round affine warps")
        out = {
            "metric": "Mvoxels/s through decon+decode, 22-bit codebook, 2048^2x64 tile",
            "value": round(value, 4),
            "unit": "Mvoxels/s",
            "n_gpus": world,
            "steps": steps_done,
            "warmup": warm_done,
            "steps_requested": args.steps,
            "warmup_requested": args.warmup,
            "wall_budget_s": budget,
            "ms_per_step": round(elapsed / max(steps_done, 1) * 1e3, 2),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"configs[{CONFIG_INDEX[args.workload]}]: synthetic {shape[0]}x{shape[1]}x{shape[2]} "
                             + (f"tiles, a {args.tiles}-tile volume dealt to {world} rank(s) in contiguous ranges, "
                                if strong else "tile per GPU, ")
                             + f"{n_bits}-bit/{len(bits_matrix)}-codeword codebook, per tile {what}"),
                "tiles_per_step": tiles_done // max(steps_done, 1),
                "tiles_timed": tiles_done,
                "emitters_per_tile": n_emitters,
            },
            "rlgc_iterations_total": total_iters,
            "mvoxel_iterations_per_s": round(total_iters * n_tile / elapsed / 1e6, 2),
            "transcripts_rank0": stats["transcripts"],
            "roofline": {
                "kernel": ("one RLGC iteration = 4 x fft_conv (k_x_forward, k_y_pass, k_z_conv, k_y_pass, k_x_inverse); the "
                           "photon split, the consensus and the update arithmetic run on the registers of three of the "
                           "X inverse passes, which also start the next convolution's X forward transform "
                           "+ 3 small finalize kernels"),
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
                "launches": passes["conv_calls"], "sampled_on": "the untimed first tile",
                "algorithmic_bytes_per_launch": RLGC_ALGORITHMIC_BYTES_PER_CONV_VOXEL * np_full,
            },
        }
        if stats["fused"] and passes["conv_calls"]:
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
            if passes["kernel_ms"][0] < 0.05 * passes["kernel_ms"][4]:
                # round 3: the split stage sits in the Hu convolution's X inverse pass and feeds the ratio convolutions' X
                # forward transforms (no stand-alone X forward launches left): that pass reads Hu's core rows and the
                # uint16 image and writes the two ratio spectra's core rows
                names = ("k_x_forward (first iteration only)", "k_y_pass<fwd>", "k_z_conv", "k_y_pass<inv>",
                         "k_x_inverse (split chain, dual, update)")
                x_split = 4.0 * rows + 2.0 * vox + 8.0 * rows
                moved = (0.0, moved[1], moved[2], moved[3], (x_split + x_dual + x_update) / 4)
            out["roofline_kernels"] = [
                {"kernel": nme, "avg_ms": round(ms / passes["conv_calls"], 3),
                 "hbm_gbs": round(bpv * np_full / (ms / passes["conv_calls"] * 1e-3) / 1e9, 1) if ms and bpv else None,
                 "bytes_per_voxel": round(bpv, 2)}
                for nme, ms, bpv in zip(names, passes["kernel_ms"], moved)]
        if stats.get("phase_s"):
            out["phase_seconds_per_tile_rank0"] = {k: round(v / max(tiles_done_rank, 1), 3) for k, v in stats["phase_s"].items()}
        if strong and stats["reg_calls"]:
            out["registration"] = {
                "what": "register_pair_to_fixed per moving round (host wall time around the call, first-seen "
                        "transform shapes included)",
                "calls_rank0": stats["reg_calls"], "avg_s": round(stats["reg_s"] / stats["reg_calls"], 4),
                "max_s": round(stats.get("reg_max_s", 0.0), 4),
                "max_shift_error_px": round(stats["reg_err_px"], 4),
                # rounds whose found shift is further than the reference's own 0.35-px acceptance bound from the planted
                # one, summed over ALL ranks (the inherited three-pixel corner-tile case of phase_cross_correlation's
                # disambiguation registers a round an image width off about once in seven at shifts of (+-3, +-1) px)
                "all_ranks": extras.get("registration_all_ranks")}
        if strong and stats.get("sofima_calls"):
            out["sofima"] = {
                "what": "affine warp + estimate_sofima_flow_field_xyz_px per moving round (two residual passes; host "
                        "wall time; the flow is not applied: the synthetic rounds are pure translations)",
                "calls_rank0": stats["sofima_calls"], "avg_s": round(stats["sofima_s"] / stats["sofima_calls"], 4),
                "last_status": stats.get("sofima_status")}
        # sanity of the decoded table against the planted emitters (voxel units, 3-voxel radius, same
        # codeword, one-to-one): the scorer of the simulation pipeline, not a timed quantity
        table = stats.get("table")
        if table is not None and len(table) and stats.get("truth") is not None:
            truth = stats["truth"]
            f1 = scoring.calculate_F1_with_radius(
                table[["tile_z", "tile_y", "tile_x"]].to_numpy(dtype=float), table["barcode_id"].to_numpy() - 1,
                truth[:, 1:4].astype(float), truth[:, 0], 3.0)[0]
            out["f1_vs_planted_emitters"] = {k: (round(v, 4) if isinstance(v, float) else int(v)) for k, v in f1.items()}
        out.update({k: v for k, v in extras.items() if v is not None and k != "registration_all_ranks"})
        if _process_group_live():
            import torch.distributed as dist

            out["process_group"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
        if world > 1:
            out["omitted_at_n_gt_1"] = ["cpu_baseline", "roofline_decode", "io"]   # rank-0, N = 1 legs only
        if "io" in out:
            io = dict(out["io"])
            step_s = elapsed / max(tiles_done, 1)
            io["value_including_load"] = round(n_tile / (step_s + io["load_seconds_per_tile"]) / 1e6, 4)
            out["io"] = io
        return out

    # ---- untimed: the first tile deconvolves so that the per-dataset global normalisation vectors exist, and
    # ---- its duration sizes the rest of the run
    my_tiles = list(contiguous_share(args.tiles, world, rank)) if strong else None
    if strong and args.tiles < world:
        raise SystemExit(f"config3 deals whole tiles: --tiles {args.tiles} < {world} ranks")
    load_tile((100 + my_tiles[0]) if strong else (1 + rank), 0)
    if disk and not strong:
        load_tile(1 + rank, 1)
    progress("synthetic tile ready" + (f" and written to {io_root}" if disk else ""))
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    if strong:
        register_rounds(True, 0)
    deconvolve_bits(True, 0)
    fix_normalization()
    decode(0)
    if disk:
        store.flush()
        store.evict_tile(0)
    torch.cuda.synchronize(device)
    first_tile_s = time.perf_counter() - t0
    # per-pass breakdown (events around every FFT pass, ~6 500 per tile): sampled here, outside the timed region
    passes = {"conv_ms": stats["conv_ms"], "conv_calls": stats["conv_calls"], "kernel_ms": list(stats["kernel_ms"])}
    if strong and stats["reg_calls"]:   # the process's first calls: every transform shape is new to rocFFT here
        extras["registration_first_tile"] = {"calls": stats["reg_calls"],
                                             "avg_s": round(stats["reg_s"] / stats["reg_calls"], 4),
                                             "max_s": round(stats.get("reg_max_s", 0.0), 4)}
    warm_iters_per_call = stats["iters"] / max(rlgc_calls_per_tile, 1)
    warm_done = 1
    progress(f"first (untimed) tile: {first_tile_s:.1f} s; {seconds_left():.0f} s of the {budget:.0f} s budget left")

    if rank == 0 and world == 1 and not disk:
        extras["roofline_decode"] = decode_kernel_leg()
        if not args.no_cpu_baseline:
            extras["cpu_baseline"] = cpu_baseline(current["readouts"][0], psf, bits_matrix, shape, n_bits,
                                                  warm_iters_per_call, rlgc_calls_per_tile)
            progress(f"cpu_baseline: {extras['cpu_baseline']['value']:.5f} Mvoxels/s")
            extras["io"] = io_leg(current["readouts"], current["predictors"], n_bits)
            progress("io leg done")

    if strong:
        # a fixed volume: every rank walks its contiguous share once; a rank whose share does not fit the
        # budget stops early and the record says how many tiles were timed
        n_fit = steps_that_fit(len(my_tiles), first_tile_s + 0.25 * first_tile_s, seconds_left())
        n_steps = agree_min(n_fit if n_fit < len(my_tiles) else 10 ** 6, world, device)
        n_steps = min(n_steps, max(len(contiguous_share(args.tiles, world, r)) for r in range(world)))
    else:
        # further warm-up tiles: at most one (the first tile already ran every kernel ~1600 times; the second
        # one runs with the plan, OTF and normalisation vectors in their steady state), and only if it leaves
        # room for timed tiles - the budget goes to the timed region
        for _ in range(min(max(args.warmup - 1, 0), 1)):
            if seconds_left() < 4.5 * first_tile_s:
                break
            step("loop", 0, None)
            warm_done += 1
        n_steps = agree_min(steps_that_fit(args.steps, first_tile_s, seconds_left()), world, device)
    reset_stats()

    elapsed_sum = 0.0
    tiles_done_rank = 0
    steps_done = 0
    last = None
    cpu_s = 0.0
    timed_wait: dict = {}
    if disk:
        for k in list(store.io_seconds):
            store.io_seconds[k] = 0.0
        store.io_counts.clear()

    def slot_of(i: int) -> int:
        # the in-memory store holds one tile; the disk store holds the rank's share (strong scaling) or - weak scaling,
        # where every step re-runs the rank's one tile - two copies of it, so that step i + 1's read-ahead has its own
        # images to load while step i consumes its own
        if not disk:
            return 0
        return i if strong else i % 2

    def prepare_slot(i: int) -> None:
        """--store disk, untimed: the input images of timed step i on disk before any step needs them."""
        if disk and strong and 0 <= i < len(my_tiles) and slot_of(i) not in slot_truth:
            load_tile(100 + my_tiles[i], slot_of(i))

    def io_record(elapsed: float, tiles: int) -> dict:
        waits = sum(timed_wait.values())     # the main thread's waits inside the timed regions only
        per_tile = elapsed / max(tiles, 1)
        return {
            "what": ("per tile: " + (f"{n_rounds} fiducial + " if strong else "") + f"{n_bits} readout uint16 + {n_bits} "
                     "predictor float32 images read ahead, " + (f"{n_rounds} + " if strong else "") + f"{n_bits} deconvolved "
                     "uint16 volumes written behind + the transcript table, qi2lab datastore layout (OME-Zarr 0.5 / Zarr v3 / "
                     f"Blosc zstd-5 bitshuffle) under {io_root}"),
            "value_including_io": round(tiles * n_tile / max(elapsed, 1e-9) / 1e6, 4),
            "value_compute_only": round(tiles * n_tile / max(elapsed - waits, 1e-9) / 1e6, 4),
            "seconds_per_tile": round(per_tile, 3),
            "caller_wait_s_per_tile": {k: round(v / max(tiles, 1), 3) for k, v in timed_wait.items()},
            "background_s_per_tile": {k: round(v / max(tiles, 1), 3) for k, v in store.io_seconds.items()
                                      if not k.endswith("_wait")},
            "per_tile_counts": {k: round(v / max(tiles, 1), 1) for k, v in store.io_counts.items()},
            "host_cpu_s_per_tile": round(cpu_s / max(tiles, 1), 2),
            "io_threads": {"read_ahead_zstd": args.io_read_threads, "write_behind_zstd": args.io_write_threads,
                           "python_io_threads": 2},
            "unit": "Mvoxels/s",
        }

    card = CardState(CardState.hwmon_of(gpu_index)).start()   # every rank samples its own card
    for i in range(n_steps):
        if strong:
            mine = i < len(my_tiles)
            if mine and i > 0 and not disk:
                load_tile(100 + my_tiles[i])       # synthesis stays outside the timed region
        else:
            mine = True
        if disk and mine:
            prepare_slot(i)
            prepare_slot(i + 1)                    # so that step i can read ahead for step i + 1
            if i == 0:
                store.prefetch_tile_inputs(slot_of(0))   # steady state: a tile's inputs are in flight before it starts
        sync()
        w0 = dict(store.io_seconds) if disk else {}
        t0 = time.perf_counter()
        c0 = time.process_time()
        if mine:
            nxt = None
            if disk and i + 1 < n_steps and (not strong or i + 1 < len(my_tiles)):
                nxt = slot_of(i + 1)
            step("loop", slot_of(i), nxt)
            tiles_done_rank += 1
        sync()
        elapsed_sum += time.perf_counter() - t0
        cpu_s += time.process_time() - c0
        for k in (("read_wait", "write_wait") if disk else ()):
            timed_wait[k] = timed_wait.get(k, 0.0) + store.io_seconds[k] - w0.get(k, 0.0)
        steps_done += 1
        elapsed, total_iters = aggregate_over_ranks(elapsed_sum, stats["iters"], world, device)
        tiles_all = aggregate_over_ranks(0.0, tiles_done_rank, world, device)[1]
        reg_off_all = aggregate_over_ranks(0.0, stats["reg_off"], world, device)[1] if strong else 0.0
        reg_calls_all = aggregate_over_ranks(0.0, stats["reg_calls"], world, device)[1] if strong else 0.0
        extras["registration_all_ranks"] = {"calls": int(reg_calls_all), "off_by_more_than_px": MISREGISTRATION_PX,
                                            "rounds_off": int(reg_off_all)} if strong else None
        cards_all = card_state_over_ranks(card.summary(), world, device) if world > 1 or force_dist else None
        if rank == 0:
            if disk:
                extras["store"] = io_record(elapsed, int(tiles_all))
            extras["card_state_rank0"] = card.summary()
            extras["card_state_all_ranks"] = cards_all
            last = record(steps_done, warm_done, elapsed, total_iters, int(tiles_all))
            print(json.dumps(last), flush=True)
        # the measured step time replaces the warm-up estimate; stop (on every rank) if the next step would
        # overrun the budget
        per_step = elapsed / steps_done
        stop = agree_min(0 if seconds_left() < 1.1 * per_step else 1, world, device) == 0
        if stop and i + 1 < n_steps:
            progress(f"stopping after {steps_done} timed step(s): wall budget")
            break
    card.stop()
    if disk:
        # the writes still queued behind the last tile belong to the job: drain them inside the clock
        sync()
        t0 = time.perf_counter()
        c0 = time.process_time()
        store.flush()
        sync()
        elapsed_sum += time.perf_counter() - t0
        timed_wait["write_wait"] = timed_wait.get("write_wait", 0.0) + time.perf_counter() - t0
        cpu_s += time.process_time() - c0
        elapsed, total_iters = aggregate_over_ranks(elapsed_sum, stats["iters"], world, device)
        tiles_all = aggregate_over_ranks(0.0, tiles_done_rank, world, device)[1]
        if rank == 0:
            extras["store"] = io_record(elapsed, int(tiles_all))
            extras["store"]["final_drain_s"] = round(time.perf_counter() - t0, 3)
            kept = stats.get("kept_u16")
            if kept is not None:       # what is on disk is what was computed, byte for byte
                back = store._io.read_image(store._readout_decon.path(store._key(kept[0], bit=store.bit_ids[0])),
                                            device=device)
                extras["store"]["round_trip_equal"] = bool(torch.equal(back, kept[1]))
            last = record(steps_done, warm_done, elapsed, total_iters, int(tiles_all))
            print(json.dumps(last), flush=True)
        import shutil

        shutil.rmtree(io_root, ignore_errors=True)
    if world > 1 or force_dist:
        import torch.distributed as dist

        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
