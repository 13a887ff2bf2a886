"""Scratch probe (not a test): board power and shader clock of the card under each kind of work of the path, each
looped for a few seconds (the card's hwmon files, as bench.py's CardState reads them).  Which part of the RLGC iteration
is it that holds the card at its power limit?  usage: gpu_probe_power.py [seconds per leg]"""
import importlib
import sys
import time

import torch

ROOT = __import__("pathlib").Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
argv, sys.argv = sys.argv, ["bench.py"]
import bench  # noqa: E402  (CardState)

sys.argv = argv
rl = importlib.import_module("merfish3d-analysis_amd.utils.rlgc")
kernels = importlib.import_module("merfish3d-analysis_amd.kernels")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
dev = torch.device("cuda:0")
hw = bench.CardState.hwmon_of(0)
print("hwmon", hw, flush=True)
g = torch.Generator(device=dev).manual_seed(0)
Z, Y, X = 64, 2048, 2048
n_em = Z * Y * X // 3400
vol = torch.zeros((Z, Y, X), device=dev)
idx = torch.randint(0, Z * Y * X, (n_em,), device=dev, generator=g)
vol.view(-1).index_put_((idx,), (torch.rand(n_em, device=dev, generator=g) * 600 + 300) / 0.0138, accumulate=True)
kernels.gaussian_lowpass(vol, (1.8, 1.6, 1.6), True, out=vol)
img = torch.poisson(vol + 100.0, generator=g).clamp_(0, 65535).to(torch.uint16)
del vol
zz, yy, xx = torch.meshgrid(*[torch.arange(n, device=dev) - n // 2 for n in (64, 51, 51)], indexing="ij")
psf = torch.exp(-0.5 * ((zz / 3.0) ** 2 + (yy / 1.8) ** 2 + (xx / 1.8) ** 2))
psf = (psf / psf.sum()).float()
out = torch.empty((Z, Y, X), dtype=torch.float32, device=dev)


def leg(name, body, unit_bytes=None):
    """body() enqueues one unit of work; looped for `seconds`, the first 1.5 s of samples dropped."""
    body()
    torch.cuda.synchronize()
    card = bench.CardState(hw, period_s=0.25).start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            body()
        torch.cuda.synchronize()
        n += 4
    dt = time.perf_counter() - t0
    card.stop()
    skip = 6
    s = {k: v[skip:] for k, v in card.samples.items() if len(v) > skip}
    mean = lambda k: sum(s[k]) / len(s[k]) if k in s else float("nan")
    rate = f"  {unit_bytes * n / dt / 1e12:5.2f} TB/s" if unit_bytes else ""
    print(f"{name:58s} {dt / n * 1e3:9.3f} ms/unit  power {mean('power_w'):6.0f} W  sclk {mean('sclk_mhz'):5.0f} MHz  "
          f"hbm {mean('hbm_c'):4.0f} C{rate}", flush=True)
    time.sleep(1.0)


big = torch.empty(612_220_032, dtype=torch.float32, device=dev)   # one padded volume, 2.45 GB
big2 = torch.empty_like(big)
big.zero_()
leg("float copy of one padded volume (2.45 GB read + write)", lambda: big2.copy_(big), 2 * big.numel() * 4)
leg("fill (write only)", lambda: big2.fill_(1.0), big.numel() * 4)
leg("sum (read only)", lambda: big.sum(), big.numel() * 4)
del big2
padded = (128, 2187, 2187)
psf_p = rl.pad_psf(psf, padded)
H = rl.transfer_function(psf_p)
vol = big.view(padded)
vol.uniform_(0.5, 1.5, generator=g)
leg("fft_conv of a padded volume (X, Y, Z, Y, X passes, no tails)", lambda: rl.fft_conv(vol, H))
leg("the photon split alone (Philox + table draws, 268 Mvoxels)", lambda: kernels.rlgc_debug_split(img, 7, 1))
del H, psf_p
rl.clear_rlgc_caches(clear_memory_pool=True)
leg("RLGC, 12 iterations per unit", lambda: rl.rlgc_device(img, psf, max_iterations=12, out=out, rng_seed=1))
a = torch.randn(8192, 8192, device=dev)
b = torch.randn(8192, 8192, device=dev)
leg("fp32 GEMM 8192^3 (rocBLAS), for scale", lambda: torch.mm(a, b))
