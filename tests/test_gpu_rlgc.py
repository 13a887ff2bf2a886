"""GPU parity: HIP RLGC (through the C ABI) against the NumPy oracle.

Tolerances: the in-LDS transforms and pocketfft differ at float32 rounding (~1e-7 per transform)
and RLGC multiplies ~25-80 updates together; measured differences are 3e-6 of the peak, the
tests allow 1e-4 (north_star: float distances within 1e-4, integers exact); iteration counts,
stop reasons and the Philox split are compared exactly.
"""

import warnings

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import philox, rlgc_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


VOLUME_TOL = 1e-4   # of the volume's peak


def _rel_err(a, b):
    return float(np.abs(a - b).max() / max(float(np.abs(b).max()), 1e-12))


def test_philox_split_is_bit_identical_to_oracle():
    """Every branch of the bounded-work sampler (csrc/m3d_binom.h): pop-count only (n < 64), q = 1, table rows from
    the smallest to the largest, row edges, and - float images only - the chunked draw above 65535 counts."""
    kernels = pkg("kernels")
    rng = np.random.default_rng(0)
    img = rng.integers(0, 3000, size=(5, 33, 47)).astype(np.uint16)
    img[0, 0, :12] = (0, 1, 63, 64, 65, 127, 128, 129, 191, 192, 65472, 65535)
    img[1] = rng.integers(0, 65536, size=img[1].shape)          # every table row
    img[2, :8] = rng.integers(60000, 65536, size=img[2, :8].shape)   # a saturated camera
    for it, seed in ((0, 42), (9, 12345678901234)):
        got = kernels.rlgc_debug_split(torch.from_numpy(img).to(DEV), seed, it).cpu().numpy()
        ref = philox.binomial_half(img.astype(np.int64), np.arange(img.size), it, seed)
        assert np.array_equal(got.ravel(), ref.astype(np.float32))
    imgf = (img.astype(np.float32) + 0.75)
    imgf[3] = rng.integers(65536, 1 << 20, size=imgf[3].shape)  # chunks of 65472 + remainder
    imgf[4, 0, :4] = (65472.0 * 2, 65472.0 * 2 + 1, 1048576.0, 3e6)   # the last is clamped to 2^20 for the draw
    got = kernels.rlgc_debug_split(torch.from_numpy(imgf).to(DEV), 42, 3).cpu().numpy()
    ref = philox.binomial_half(np.minimum(imgf, 1048576.0).astype(np.int64), np.arange(img.size), 3, 42)
    assert np.array_equal(got.ravel(), ref.astype(np.float32))


def test_table_draws_on_and_around_every_threshold_land_on_the_exact_count():
    """The device search starts from a float quantile guess (v_log_f32 / v_sqrt_f32 / v_rcp_f32: not the host's
    arithmetic) and is decided by the integer thresholds: uniforms placed on, one below and one above every
    threshold of eight rows must give the oracle's count."""
    from test_binomial_split import threshold_probes

    kernels = pkg("kernels")
    q, hi, lo, want = threshold_probes(np.random.default_rng(3), (2, 3, 7, 40, 200, 511, 937, 1023), n_random=20000)
    to = lambda a: torch.from_numpy(a.view(np.int32)).to(DEV).view(torch.uint32)  # noqa: E731
    got = kernels.binom_debug_table_draw(to(q), to(hi), to(lo)).view(torch.int32).cpu().numpy()
    assert np.array_equal(got.astype(np.int64), want)


def test_split_of_a_bright_volume_has_binomial_moments():
    """Size-independent property on 2^23 voxels of 60000 counts: mean n / 2 and variance n / 4 of the device draws
    (the oracle-identical stream, checked on the device so that no table entry is out of place in HBM)."""
    kernels = pkg("kernels")
    n = 60000
    img = torch.full((32, 512, 512), n, dtype=torch.uint16, device=DEV)
    s = kernels.rlgc_debug_split(img, 7, 1).double()
    count = s.numel()
    assert abs(float(s.mean()) - n / 2) < 5 * (n / 4 / count) ** 0.5
    assert abs(float(s.var()) / (n / 4) - 1.0) < 5 * (2.0 / count) ** 0.5
    assert float(s.min()) > n / 2 - 7 * (n / 4) ** 0.5 and float(s.max()) < n / 2 + 7 * (n / 4) ** 0.5


@pytest.mark.parametrize("name", ["rlgc_a", "rlgc_b"])
def test_rlgc_matches_oracle_with_philox_split(golden, name):
    rl = pkg("utils.rlgc")
    g = golden(name)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref, rinfo = rlgc_ref.rlgc(g["image"], g["psf"], rng="philox", return_info=True)
    out, info = rl.rlgc(g["image"], g["psf"], gpu_id=0, return_info=True, release_memory=False)
    assert out.dtype == np.float32 and out.shape == g["image"].shape
    assert info.padded_shape == rinfo.padded_shape
    assert (info.iterations, info.stop_reason) == (rinfo.iterations, rinfo.stop_reason)
    np.testing.assert_allclose(info.kld_split1, rinfo.kld_split1, rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(info.kld_split2, rinfo.kld_split2, rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(info.updated_fraction, rinfo.updated_fraction, atol=1e-4)
    assert _rel_err(out, ref) < VOLUME_TOL
    # same result when the image arrives as float32 or already sits in HBM
    out_f, _ = rl.rlgc(g["image"].astype(np.float32), g["psf"], return_info=True, release_memory=False)
    assert np.array_equal(out_f, out)
    out_t = rl.rlgc(torch.from_numpy(g["image"]).to(DEV), torch.from_numpy(g["psf"]).to(DEV),
                    release_memory=False)
    assert isinstance(out_t, torch.Tensor) and np.array_equal(out_t.cpu().numpy(), out)


def test_rlgc_matches_reference_golden_with_injected_reference_splits(golden):
    """Feed the exact binomial draws the reference run used (NumPy stream) through the test
    hook: the HIP solver must then land on the reference's own output."""
    rl = pkg("utils.rlgc")
    g = golden("rlgc_a")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref, rinfo = rlgc_ref.rlgc(g["image"], g["psf"], rng="numpy", return_info=True,
                                   record_splits=True)
    assert np.array_equal(ref, g["output"])  # oracle == reference (also a CPU test)
    pads = rinfo.pad_width
    splits = np.stack([rlgc_ref.remove_padding_zyx(s, pads) for s in rinfo.splits])
    out, info = rl.rlgc_device(torch.from_numpy(g["image"]).to(DEV), torch.from_numpy(g["psf"]).to(DEV),
                               injected_splits=torch.from_numpy(splits).to(DEV))
    assert (info.iterations, info.stop_reason) == (rinfo.iterations, rinfo.stop_reason)
    assert _rel_err(out.cpu().numpy(), g["output"]) < VOLUME_TOL


def test_rlgc_tiled_route_and_2d_input(golden):
    rl = pkg("utils.rlgc")
    g = golden("rlgc_tiled")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = rlgc_ref.chunked_rlgc(g["image"], g["psf"], crop_yx=int(g["crop_yx"]),
                                    rng_seed=int(g["rng_seed"]), rng="philox")
    out = rl.chunked_rlgc(g["image"], g["psf"], crop_yx=int(g["crop_yx"]), rng_seed=int(g["rng_seed"]))
    assert out.shape == ref.shape and _rel_err(out, ref) < VOLUME_TOL
    g2 = golden("rlgc_2d")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref2 = rlgc_ref.chunked_rlgc(g2["image"], g2["psf"], rng="philox")
    out2 = rl.chunked_rlgc(g2["image"], g2["psf"])
    assert out2.shape == g2["image"].shape and _rel_err(out2, ref2) < VOLUME_TOL


def _phantom(shape, psf_shape, sigma, n_spots, seed):
    rng = np.random.default_rng(seed)
    zz, yy, xx = np.meshgrid(*[np.arange(n, dtype=np.float64) - n // 2 for n in psf_shape], indexing="ij")
    psf = np.exp(-0.5 * ((zz / sigma[0]) ** 2 + (yy / sigma[1]) ** 2 + (xx / sigma[2]) ** 2))
    psf = (psf / psf.sum()).astype(np.float32)
    vol = np.zeros(shape)
    idx = tuple(rng.integers(0, s, size=n_spots) for s in shape)
    np.add.at(vol, idx, rng.uniform(400, 1500, size=n_spots))
    from scipy.ndimage import convolve

    img = rng.poisson(convolve(vol, psf.astype(np.float64), mode="reflect") * 40.0 + 20.0)
    return np.clip(img, 0, 65535).astype(np.uint16), psf


@pytest.mark.parametrize("shape,psf_shape,padded", [((22, 70, 72), (9, 9, 9), (32, 81, 81)),
                                                    ((52, 70, 150), (11, 9, 9), (64, 81, 162)),
                                                    ((100, 70, 72), (17, 9, 9), (128, 81, 81))])
def test_solver_with_register_z_pass_matches_oracle(shape, psf_shape, padded):
    """Padded Z = 32 / 64 / 128 (the last is the headline configuration's variant: 64 columns per block, the two
    ratio convolutions in one paired launch): the production Z pass (k_z_conv_reg: 16-point DFTs in registers around one LDS exchange),
    the planes mirrored in LDS, the fused consensus / update tails and the register-edge X passes (81 = 9 * 9,
    162 = 9 * 2 * 9) all inside the solver, against the independent NumPy implementation - same Philox stream."""
    rl = pkg("utils.rlgc")
    img, psf = _phantom(shape, psf_shape, (1.6, 1.5, 1.5), 90, seed=13)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref, rinfo = rlgc_ref.rlgc(img, psf, rng="philox", return_info=True)
    assert tuple(rinfo.padded_shape) == padded
    out, info = rl.rlgc(img, psf, gpu_id=0, return_info=True, release_memory=False)
    assert tuple(info.padded_shape) == padded
    assert (info.iterations, info.stop_reason) == (rinfo.iterations, rinfo.stop_reason)
    np.testing.assert_allclose(info.kld_split1, rinfo.kld_split1, rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(info.updated_fraction, rinfo.updated_fraction, atol=1e-4)
    assert _rel_err(out, ref) < VOLUME_TOL
    # what DataRegistration stores: clip -> uint16 (truncation).  FFT rounding moves a value across an integer in a
    # small fraction of voxels, never by more than one count
    a, b = np.clip(out, 0, 65535).astype(np.uint16).astype(np.int64), np.clip(ref, 0, 65535).astype(np.uint16).astype(np.int64)
    assert np.abs(a - b).max() <= 1 and np.mean(a != b) < 5e-3, (np.abs(a - b).max(), np.mean(a != b))
    rl.clear_rlgc_caches(clear_memory_pool=True)


@pytest.mark.parametrize("shape,padded", [((22, 70, 56), (32, 81, 64)),        # 64 = 8 * 8
                                          ((22, 70, 120), (32, 81, 128)),      # 128 = 8 * 2 * 8
                                          ((22, 40, 184), (32, 48, 192)),      # 192 = 8 * 3 * 8
                                          ((22, 30, 568), (32, 48, 576)),      # 576 = 8 * 9 * 8
                                          ((22, 28, 1144), (32, 36, 1152)),    # 1152 = 8 * 9 * 2 * 8
                                          ((22, 27, 2040), (32, 36, 2048)),    # 2048 = 8 * 8 * 4 * 8: a fifth wave
                                          ((22, 27, 2584), (32, 36, 2592)),    # 2592 = 9 * 8 * 4 * 9: radix-9 edges, 288
                                                                               # edge items - a fifth wave with elements
                                          ((22, 27, 640), (32, 36, 648))])     # 648 = 9 * 8 * 9: radix 8 between radix-9 edges
def test_solver_with_radix8_register_edges_matches_oracle(shape, padded):
    """Padded X lengths 64 m without two radix-9 stages run the register-edge X passes with radix-8 edge stages (round
    3): forward (mirrored and pruned inputs), plain / update inverse, the dual pass with its chained forward transform,
    whole-line staging - all inside the solver, against the independent NumPy implementation, same Philox stream.  At
    2048 a packed row has 129 block pairs, so two rows need a fifth wave per workgroup."""
    rl = pkg("utils.rlgc")
    img, psf = _phantom(shape, (9, 9, 9), (1.6, 1.5, 1.5), 90, seed=29)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref, rinfo = rlgc_ref.rlgc(img, psf, rng="philox", return_info=True)
    assert tuple(rinfo.padded_shape) == padded
    out, info = rl.rlgc(img, psf, gpu_id=0, return_info=True, release_memory=False)
    assert tuple(info.padded_shape) == padded
    assert (info.iterations, info.stop_reason) == (rinfo.iterations, rinfo.stop_reason)
    np.testing.assert_allclose(info.kld_split1, rinfo.kld_split1, rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(info.updated_fraction, rinfo.updated_fraction, atol=1e-4)
    assert _rel_err(out, ref) < VOLUME_TOL
    rl.clear_rlgc_caches(clear_memory_pool=True)


def test_rlgc_properties_at_medium_size():
    """Size-independent properties on a 32x256x256 volume (oracle too slow to be the checker)."""
    rl = pkg("utils.rlgc")
    gen = torch.Generator(device=DEV).manual_seed(7)
    z, y, x = 32, 256, 256
    truth = torch.zeros((z, y, x), device=DEV)
    idx = torch.randint(0, z * y * x, (400,), device=DEV, generator=gen)
    truth.view(-1)[idx] = torch.rand(400, device=DEV, generator=gen) * 4000 + 1000
    zz, yy, xx = torch.meshgrid(*[torch.arange(n, device=DEV) - n // 2 for n in (15, 17, 17)], indexing="ij")
    psf = torch.exp(-0.5 * ((zz / 2.0) ** 2 + (yy / 1.7) ** 2 + (xx / 1.7) ** 2))
    psf = psf / psf.sum()
    blurred = torch.nn.functional.conv3d(truth[None, None], psf[None, None], padding="same")[0, 0]
    img = torch.poisson(blurred + 30.0, generator=gen).clamp(0, 65535)
    img_u16 = img.to(torch.uint16)
    out1, i1 = rl.rlgc_device(img_u16, psf, rng_seed=42)
    out2, i2 = rl.rlgc_device(img_u16, psf, rng_seed=42)
    out3, i3 = rl.rlgc_device(img_u16, psf, rng_seed=43)
    assert torch.equal(out1, out2) and i1.iterations == i2.iterations  # bit-reproducible
    assert not torch.equal(out1, out3)                                 # the seed matters
    assert torch.isfinite(out1).all() and float(out1.min()) >= 0.0
    flux = float(out1.double().sum() / img.double().sum())
    assert abs(flux - 2.0) < 0.02, flux                                # reference flux doubling
    assert 2 <= i1.iterations <= 200 and i1.stop_reason in ("restore_previous_recon", "limit", "max_delta")
    # sharper than the input: the brightest emitter voxel gains contrast
    assert float(out1.max()) > 2.0 * float(img.max())
    rl.clear_rlgc_caches(clear_memory_pool=True)


def test_rlgc_argument_errors():
    rl = pkg("utils.rlgc")
    with pytest.raises(ValueError):
        rl.chunked_rlgc(np.zeros((4, 8, 8), np.uint16), np.ones((3, 3, 3), np.float32), crop_z=4)
    with pytest.raises(ValueError):
        rl.chunked_rlgc(np.zeros((2, 4, 8, 8), np.uint16), np.ones((3, 3, 3), np.float32))
    with pytest.raises(ValueError):
        rl._chunked_rlgc_once(np.zeros((4, 8, 8), np.uint16), np.ones((3, 3, 3), np.float32), crop_yx=0)


def test_clip_to_u16():
    kernels = pkg("kernels")
    v = torch.tensor([-5.0, 0.0, 0.99, 1.5, 65534.7, 65535.0, 70000.0, 1e9], device=DEV)
    got = kernels.clip_to_u16(v).cpu().numpy()
    ref = np.clip(v.cpu().numpy(), 0, 65535).astype(np.uint16)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("shape,psf_shape", [((12, 40, 44), (7, 9, 9)), ((16, 64, 64), (9, 11, 11)),
                                             ((5, 30, 100), (5, 7, 13)), ((1, 48, 48), (1, 9, 9))])
def test_fused_fft_engine_matches_rocfft(shape, psf_shape):
    """The in-LDS convolution engine against the rocFFT path on the same solver."""
    rl = pkg("utils.rlgc")
    rng = np.random.default_rng(5)
    img = rng.poisson(rng.gamma(2.0, 60.0, size=shape)).astype(np.uint16)
    zz, yy, xx = np.meshgrid(*[np.arange(n) - n // 2 for n in psf_shape], indexing="ij")
    psf = np.exp(-0.5 * ((zz / 1.5) ** 2 + (yy / 1.7) ** 2 + (xx / 1.7) ** 2)).astype(np.float32)
    try:
        rl.set_fft_engine("rocfft")
        ref, rinfo = rl.rlgc(img, psf, return_info=True, release_memory=False, max_iterations=12)
        rl.set_fft_engine("fused")
        out, info = rl.rlgc(img, psf, return_info=True, release_memory=False, max_iterations=12)
    finally:
        rl.set_fft_engine("fused")
    assert info.fused_engine and not rinfo.fused_engine
    assert (info.iterations, info.stop_reason) == (rinfo.iterations, rinfo.stop_reason)
    np.testing.assert_allclose(info.kld_split1, rinfo.kld_split1, rtol=1e-4, atol=1e-7)
    assert _rel_err(out, ref) < 2e-4


@pytest.mark.parametrize("shape,psf_shape", [((12, 40, 44), (7, 9, 9)), ((9, 33, 70), (5, 8, 13)), ((1, 48, 48), (1, 9, 9))])
def test_support_pruned_convolutions_equal_the_full_ones(shape, psf_shape, monkeypatch):
    """Skipping rows / planes that are zero on input or unread on output (ConvBox) must not change
    the solver: same iterations, same traces, same volume (row pairs that straddle the box edge
    see a zero partner instead of its true value - float32 rounding only)."""
    rl = pkg("utils.rlgc")
    rng = np.random.default_rng(11)
    img = rng.poisson(rng.gamma(2.0, 60.0, size=shape)).astype(np.uint16)
    zz, yy, xx = np.meshgrid(*[np.arange(n) - n // 2 for n in psf_shape], indexing="ij")
    psf = np.exp(-0.5 * ((zz / 1.5) ** 2 + (yy / 1.7) ** 2 + (xx / 1.7) ** 2)).astype(np.float32)
    monkeypatch.setenv("M3D_FFT_PRUNE", "0")
    ref, rinfo = rl.rlgc(img, psf, return_info=True, release_memory=False, max_iterations=15)
    monkeypatch.setenv("M3D_FFT_PRUNE", "1")
    out, info = rl.rlgc(img, psf, return_info=True, release_memory=False, max_iterations=15)
    assert info.fused_engine and rinfo.fused_engine
    assert (info.iterations, info.stop_reason) == (rinfo.iterations, rinfo.stop_reason)
    np.testing.assert_allclose(info.kld_split1, rinfo.kld_split1, rtol=1e-5, atol=1e-8)
    assert _rel_err(out, ref) < 2e-5


def test_chained_passes_change_nothing(monkeypatch):
    """Lateral size 2187 (the only one with the chained builds).  (a) The update pass that writes the next iteration's X
    forward transform over the spectrum rows it has read (M3D_RLGC_CHAIN_UPDATE, default on) and (b) the split stage run
    on the registers of the Hu convolution's X inverse pass and handed to the ratio convolutions' X forward transforms
    (M3D_RLGC_CHAIN_SPLIT, default on: Hu and both ratio volumes never stored) against the unchained sequence - same
    iterations, same stop, same KL traces, the same volume to the float32 rounding of a transform; also with injected
    splits (the parity hook goes through the fused kernel too)."""
    rl = pkg("utils.rlgc")
    gen = torch.Generator(device=DEV).manual_seed(3)
    shape = (22, 2140, 2140)       # + the PSF halo and FFT smoothing = 2187 laterally, 32 planes (register Z pass)
    img = torch.poisson(torch.rand(shape, device=DEV, generator=gen) * 150 + 40, generator=gen).to(torch.uint16)
    img[3, 100:140, 200:260] = 40000          # a saturated patch: table rows of the split's upper range
    zz, yy, xx = torch.meshgrid(*[torch.arange(n, device=DEV) - n // 2 for n in (9, 47, 47)], indexing="ij")
    psf = torch.exp(-0.5 * ((zz / 1.5) ** 2 + (yy / 1.8) ** 2 + (xx / 1.8) ** 2)).float()
    out = {}
    for update, split in (("0", "0"), ("1", "0"), ("1", "1"), ("0", "1")):
        monkeypatch.setenv("M3D_RLGC_CHAIN_UPDATE", update)
        monkeypatch.setenv("M3D_RLGC_CHAIN_SPLIT", split)
        o, info = rl.rlgc_device(img, psf, max_iterations=6, rng_seed=11)
        out[update + split] = (o.clone(), info)
    base = out["00"]
    assert tuple(base[1].padded_shape) == (32, 2187, 2187), base[1].padded_shape
    for key in ("10", "11", "01"):
        a = out[key]
        assert (a[1].iterations, a[1].stop_reason) == (base[1].iterations, base[1].stop_reason), key
        np.testing.assert_allclose(a[1].kld_split1, base[1].kld_split1, rtol=1e-5, atol=1e-8, err_msg=key)
        np.testing.assert_allclose(a[1].kld_image, base[1].kld_image, rtol=1e-5, atol=1e-8, err_msg=key)
        err = float((a[0] - base[0]).abs().max() / base[0].abs().max())
        assert err < 2e-5, (key, err)
    # a float32 image takes the kernel's other instantiation (chunked draw above 65535 counts): same volume as the
    # uint16 image where the counts agree, and a count beyond uint16 goes through fused and unfused alike
    imgf = img.float()
    monkeypatch.setenv("M3D_RLGC_CHAIN_SPLIT", "1")
    of, _ = rl.rlgc_device(imgf, psf, max_iterations=6, rng_seed=11)
    assert torch.equal(of, out["11"][0])
    imgf[5, 700:720, 900:930] = 150000.0
    big = {}
    for split in ("0", "1"):
        monkeypatch.setenv("M3D_RLGC_CHAIN_SPLIT", split)
        o, info = rl.rlgc_device(imgf, psf, max_iterations=3, rng_seed=11)
        big[split] = (o.clone(), info)
    assert big["0"][1].iterations == big["1"][1].iterations
    assert float((big["1"][0] - big["0"][0]).abs().max() / big["0"][0].abs().max()) < 2e-5
    # injected splits: two iterations' worth of a fixed half / half split
    inj = torch.stack([torch.floor(img.float() / 2), torch.ceil(img.float() / 2)])
    res = {}
    for split in ("0", "1"):
        monkeypatch.setenv("M3D_RLGC_CHAIN_SPLIT", split)
        o, info = rl.rlgc_device(img, psf, max_iterations=2, rng_seed=11, injected_splits=inj)
        res[split] = (o.clone(), info)
    assert res["0"][1].iterations == res["1"][1].iterations
    np.testing.assert_allclose(res["1"][1].kld_split2, res["0"][1].kld_split2, rtol=1e-5, atol=1e-8)
    assert float((res["1"][0] - res["0"][0]).abs().max() / res["0"][0].abs().max()) < 2e-5
    rl.clear_rlgc_caches(clear_memory_pool=True)


def test_static_geometry_kernels_equal_the_generic_ones(monkeypatch):
    """Lateral size 2187 runs kernels compiled for that geometry (register-resident twiddles, unrolled middle stages,
    static Y stages, chained tails); M3D_FFT_XSTATIC=0 / M3D_FFT_YSTATIC=0 send the same solve through the run-time
    geometry kernels that the oracle comparisons at other sizes pin: same iterations, KL traces and volume to the
    float32 rounding of a transform."""
    rl = pkg("utils.rlgc")
    gen = torch.Generator(device=DEV).manual_seed(17)
    shape = (22, 2140, 2140)
    img = torch.poisson(torch.rand(shape, device=DEV, generator=gen) * 300 + 60, generator=gen).to(torch.uint16)
    zz, yy, xx = torch.meshgrid(*[torch.arange(n, device=DEV) - n // 2 for n in (9, 47, 47)], indexing="ij")
    psf = torch.exp(-0.5 * ((zz / 1.5) ** 2 + (yy / 1.8) ** 2 + (xx / 1.8) ** 2)).float()
    out = {}
    for static in ("1", "0"):
        rl.clear_rlgc_caches(clear_memory_pool=True)        # the switches are read when a plan is created
        monkeypatch.setenv("M3D_FFT_XSTATIC", static)
        monkeypatch.setenv("M3D_FFT_YSTATIC", static)
        o, info = rl.rlgc_device(img, psf, max_iterations=5, rng_seed=4)
        out[static] = (o.clone(), info)
    rl.clear_rlgc_caches(clear_memory_pool=True)
    a, b = out["1"], out["0"]
    assert tuple(a[1].padded_shape) == (32, 2187, 2187)
    assert (a[1].iterations, a[1].stop_reason) == (b[1].iterations, b[1].stop_reason)
    np.testing.assert_allclose(a[1].kld_split1, b[1].kld_split1, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(a[1].kld_image, b[1].kld_image, rtol=1e-5, atol=1e-8)
    assert float((a[0] - b[0]).abs().max() / b[0].abs().max()) < 2e-5


@pytest.mark.parametrize("ly,yp", [(2002, 2048), (1680, 1728), (1250, 1296), (1100, 1152),
                                   (2546, 2592),      # tiles of two columns (9 9 8 4 at 640 threads)
                                   (530, 576)])       # three stages (9 8 8)
def test_static_y_kernels_of_other_lengths_equal_the_generic_ones(monkeypatch, ly, yp):
    """Padded Y lengths 2048 / 1728 / 1296 / 1152 run static Y kernels with their own radix lists (8 8 8 4, 9 8 8 3,
    9 9 4 4, 9 8 4 4: forward, inverse with cropped output, the mirrored forward pass); M3D_FFT_YSTATIC=0 sends the same
    solve through the run-time kernels (radix list 4 4 4 4 4 2 for 2048, ...) that the oracle comparisons pin: same
    iterations, KL traces and volume to the float32 rounding of a transform."""
    rl = pkg("utils.rlgc")
    gen = torch.Generator(device=DEV).manual_seed(ly)
    shape = (22, ly, 70)
    img = torch.poisson(torch.rand(shape, device=DEV, generator=gen) * 300 + 60, generator=gen).to(torch.uint16)
    zz, yy, xx = torch.meshgrid(*[torch.arange(n, device=DEV) - n // 2 for n in (9, 47, 9)], indexing="ij")
    psf = torch.exp(-0.5 * ((zz / 1.5) ** 2 + (yy / 1.8) ** 2 + (xx / 1.4) ** 2)).float()
    out = {}
    for static in ("1", "0"):
        rl.clear_rlgc_caches(clear_memory_pool=True)        # the switch is read when a plan is created
        monkeypatch.setenv("M3D_FFT_YSTATIC", static)
        o, info = rl.rlgc_device(img, psf, max_iterations=5, rng_seed=4)
        out[static] = (o.clone(), info)
    rl.clear_rlgc_caches(clear_memory_pool=True)
    a, b = out["1"], out["0"]
    assert tuple(a[1].padded_shape) == (32, yp, 81)
    assert (a[1].iterations, a[1].stop_reason) == (b[1].iterations, b[1].stop_reason)
    np.testing.assert_allclose(a[1].kld_split1, b[1].kld_split1, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(a[1].kld_image, b[1].kld_image, rtol=1e-5, atol=1e-8)
    assert float((a[0] - b[0]).abs().max() / b[0].abs().max()) < 2e-5


def test_setup_is_reused_for_the_same_psf_and_rebuilt_for_another(monkeypatch):
    """OTF / update_norm stay in the plan's workspace between runs (a tile's bits share their PSF): reusing
    them must give the bits a fresh setup gives, and a different PSF must not pick up stale volumes."""
    rl = pkg("utils.rlgc")
    rng = np.random.default_rng(23)
    shape = (10, 48, 52)
    imgs = [rng.poisson(rng.gamma(2.0, 60.0, size=shape)).astype(np.uint16) for _ in range(2)]
    zz, yy, xx = np.meshgrid(*[np.arange(n) - n // 2 for n in (5, 9, 9)], indexing="ij")
    psf_a = np.exp(-0.5 * ((zz / 1.2) ** 2 + (yy / 1.6) ** 2 + (xx / 1.6) ** 2)).astype(np.float32)
    psf_b = np.exp(-0.5 * ((zz / 1.5) ** 2 + (yy / 2.2) ** 2 + (xx / 2.2) ** 2)).astype(np.float32)
    monkeypatch.setenv("M3D_RLGC_NO_SETUP_CACHE", "1")
    fresh = [rl.rlgc(imgs[0], psf_a, release_memory=False, max_iterations=8),
             rl.rlgc(imgs[1], psf_a, release_memory=False, max_iterations=8),
             rl.rlgc(imgs[1], psf_b, release_memory=False, max_iterations=8),
             rl.rlgc(imgs[0], psf_a, release_memory=False, max_iterations=8, normalize_psf=False)]
    monkeypatch.setenv("M3D_RLGC_NO_SETUP_CACHE", "0")
    rl.clear_rlgc_caches()
    cached = [rl.rlgc(imgs[0], psf_a, release_memory=False, max_iterations=8),
              rl.rlgc(imgs[1], psf_a, release_memory=False, max_iterations=8),       # same PSF: setup reused
              rl.rlgc(imgs[1], psf_b, release_memory=False, max_iterations=8),       # other PSF: rebuilt
              rl.rlgc(imgs[0], psf_a, release_memory=False, max_iterations=8, normalize_psf=False)]
    for a, b in zip(fresh, cached):
        np.testing.assert_array_equal(a, b)
    assert not np.array_equal(cached[1], cached[2])
