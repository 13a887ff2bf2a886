"""GPU: BASELINE.json configs[0] end to end - a 16x256x256 tile, 16-bit codebook: RLGC deconvolution of every readout
bit (same Philox split stream), clip -> uint16 as DataRegistration stores it (DataRegistration.py:993-995), then
decode_one_tile (readout x predictor, low-pass, pixel decode, connected components, transcript table) - through the
oracle and through the HIP path, compared stage by stage.  The contract is north_star's: identical transcript tables
(integer columns and coordinates), floats within tolerance.  The solver is capped at four iterations on both sides to
keep the NumPy side to about a minute; every stage still runs at the configuration's real size."""

import warnings

import numpy as np
import pytest
import torch
from scipy.ndimage import gaussian_filter

from conftest import pkg
from helpers import codebook_frame, sort_table
from oracle import ccl_ref, decode_ref, rlgc_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPE, N_BITS, ITERS = (16, 256, 256), 16, 4


def _tile(seed=0):
    import sys

    sys.argv = ["bench.py"]
    import bench

    rng = np.random.default_rng(seed)
    cb = bench.hw4_codebook(N_BITS)
    mat = cb.iloc[:, 1:].to_numpy(dtype=np.int64)
    n = 1200
    z = rng.integers(2, SHAPE[0] - 2, n)
    y = rng.integers(4, SHAPE[1] - 4, n)
    x = rng.integers(4, SHAPE[2] - 4, n)
    code = rng.integers(0, mat.shape[0], n)
    amp = rng.uniform(300, 900, n) / 0.0138
    readouts, predictors = [], []
    for b in range(N_BITS):
        sel = mat[code, b] > 0
        vol = np.zeros(SHAPE)
        np.add.at(vol, (z[sel], y[sel], x[sel]), amp[sel] * rng.uniform(0.8, 1.2, int(sel.sum())))
        vol = gaussian_filter(vol, (1.8, 1.6, 1.6)) + 100.0
        readouts.append(np.clip(rng.poisson(vol), 0, 65535).astype(np.uint16))
        pred = np.zeros(SHAPE)
        np.add.at(pred, (z[sel], y[sel], x[sel]), 1.0)
        pred = np.clip(gaussian_filter(pred, (1.0, 1.2, 1.2)) / 0.0441 + 0.01, 0.0, 1.0)
        predictors.append(pred.astype(np.float32))
    zz, yy, xx = np.meshgrid(*[np.arange(s, dtype=np.float64) - s // 2 for s in (16, 51, 51)], indexing="ij")
    psf = np.exp(-0.5 * ((zz / 1.8) ** 2 + (yy / 1.6) ** 2 + (xx / 1.6) ** 2))
    return cb, mat, readouts, predictors, (psf / psf.sum()).astype(np.float32)


def test_config1_tile_oracle_and_hip_agree_from_raw_counts_to_transcripts():
    rl = pkg("utils.rlgc")
    kernels = pkg("kernels")
    PD = pkg("PixelDecoder")
    ds_mod = pkg("datastore")
    cb, mat, readouts, predictors, psf = _tile()
    psf_t = torch.from_numpy(psf).to(DEV)

    # ---- stage 1: deconvolution -> uint16, both paths
    rlgc_ref.FFT_WORKERS = 16
    decon_ref, decon_hip, flips, iters = [], [], 0, []
    try:
        for b in range(N_BITS):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ref, rinfo = rlgc_ref.rlgc(readouts[b], psf, rng="philox", rng_seed=42, max_iterations=ITERS,
                                           return_info=True)
            out, info = rl.rlgc_device(torch.from_numpy(readouts[b]).to(DEV), psf_t, rng_seed=42, max_iterations=ITERS)
            assert (info.iterations, info.stop_reason) == (rinfo.iterations, rinfo.stop_reason)
            iters.append(info.iterations)
            u_ref = np.clip(ref, 0, 65535).astype(np.uint16)
            u_hip = kernels.clip_to_u16(out).cpu().numpy()
            d = np.abs(u_ref.astype(np.int64) - u_hip.astype(np.int64))
            assert d.max() <= 1, (b, int(d.max()))        # a value that sits on an integer, nothing more
            flips += int((d > 0).sum())
            decon_ref.append(u_ref)
            decon_hip.append(u_hip)
    finally:
        rlgc_ref.FFT_WORKERS = 0
    n_vox = N_BITS * int(np.prod(SHAPE))
    print(f"\\nconfig 1 end to end: {sum(iters)} RLGC iterations; uint16 voxels that differ: {flips} of {n_vox} "
          f"({flips / n_vox:.2e}), all by one count")
    assert flips / n_vox < 2e-3

    # ---- stage 2: decode.  Same analysis parameters on both sides (from the oracle's low-passed stack)
    stack_ref = np.stack([decon_ref[b].astype(np.float32) * predictors[b] for b in range(N_BITS)])
    lp_ref = decode_ref.lowpass(stack_ref, (3.0, 1.0, 1.0), True)
    bkd = np.median(lp_ref.reshape(N_BITS, -1), axis=1).astype(np.float32)
    nrm = np.maximum(np.percentile(lp_ref.reshape(N_BITS, -1), 99.9, axis=1).astype(np.float32) - bkd, 1e-3)
    matrix, genes, pix_thr, tr_thr, _blank = decode_ref.load_codebook(mat, cb["gene_id"].tolist())
    cbn = decode_ref.normalize_codebook(matrix)
    mag_thr = (0.9, 10.0)
    dec_ref, mag_r, dist_r, sc_r = decode_ref.decode_pixels(lp_ref, cbn, pix_thr, mag_thr, background=bkd,
                                                            normalization=nrm, arith="device")
    table_ref = ccl_ref.extract_barcodes(dec_ref, sc_r, mag_r, dist_r, matrix, genes, tr_thr, minimum_pixels=3,
                                         spacing=(0.315, 0.098, 0.098))

    store = ds_mod.InMemoryDataStore(num_tiles=1, num_rounds=1, num_bits=N_BITS, codebook=cb,
                                     voxel_size_zyx_um=(0.315, 0.098, 0.098))
    for b in range(N_BITS):
        store.set_corrected_image(readouts[b], tile=0, bit=b)
        store.save_local_deconvolved_readout_image(decon_hip[b], tile=0, bit=b)
        store.save_local_feature_predictor_image(predictors[b], tile=0, bit=b)
    store.save_decode_normalization_vectors(None, "global", nrm, bkd)
    dec = PD.PixelDecoder(store, merfish_bits=N_BITS, verbose=0)
    res = dec.decode_one_tile(tile_idx=0, return_results=True, lowpass_sigma=(3.0, 1.0, 1.0), magnitude_threshold=mag_thr,
                              minimum_pixels=3, normalization_method="global")
    pix_same = float(np.mean(res[4] == dec_ref))
    got, want = sort_table(dec.decoded_barcodes), sort_table(table_ref)
    print(f"decoded voxels identical: {pix_same:.6f}; transcripts oracle {len(want)} / HIP {len(got)}")
    assert pix_same > 0.9999
    assert len(want) > 300
    key = ["tile_z", "tile_y", "tile_x", "gene_id", "area"]
    same = got[key].astype({"gene_id": str}).merge(want[key].astype({"gene_id": str}), on=key)
    # identical tables up to the handful of transcripts that contain a voxel sitting exactly on a threshold
    assert len(got) == len(want) and len(same) >= len(want) - 2, (len(got), len(want), len(same))
    if len(same) == len(want):
        for c in ("barcode_id", "on_bit_1", "on_bit_4"):
            assert (got[c].to_numpy() == want[c].to_numpy()).all(), c
        for c in ("global_z", "global_y", "global_x"):
            np.testing.assert_allclose(got[c].astype(float), want[c].astype(float), atol=0.0101, err_msg=c)
        np.testing.assert_allclose(got["distance_min"].astype(float), want["distance_min"].astype(float), atol=1.5e-3)
    rl.clear_rlgc_caches(clear_memory_pool=True)
