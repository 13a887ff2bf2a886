"""GPU parity: resampling kernels and global-normalisation statistics vs the reference run."""

import numpy as np
import pytest
import torch

from conftest import pkg
from helpers import codebook_frame, store_from_stack

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_affine_warp_matches_reference_run(golden):
    mv = pkg("utils.multiview_registration")
    g = golden("registration")
    T, sp, vol = g["transform"], tuple(g["spacing"]), g["volume"]
    out = mv.warp_array_to_reference_gpu(vol, transform_zyx_um=T, spacing_zyx_um=sp, reference_shape=vol.shape)
    assert out.dtype == np.float32
    np.testing.assert_allclose(out, g["warped_f32"], rtol=1e-6, atol=1e-4)
    assert np.mean(out == g["warped_f32"]) > 0.99
    out_u = mv.warp_array_to_reference_gpu(vol.astype(np.uint16), transform_zyx_um=T, spacing_zyx_um=sp,
                                           reference_shape=g["warped_u16"].shape)
    assert out_u.dtype == np.uint16 and out_u.shape == g["warped_u16"].shape
    assert np.array_equal(out_u, g["warped_u16"])
    # device in -> device out
    t = mv.warp_array_to_reference_gpu(torch.from_numpy(vol).to(DEV), transform_zyx_um=T, spacing_zyx_um=sp,
                                       reference_shape=vol.shape)
    assert isinstance(t, torch.Tensor) and np.array_equal(t.cpu().numpy(), out)
    # identity is exact
    same = mv.warp_array_to_reference_gpu(vol, transform_zyx_um=np.eye(4), spacing_zyx_um=sp, reference_shape=vol.shape)
    assert np.array_equal(same, vol)


def test_flow_warp_matches_reference_run(golden):
    mv = pkg("utils.multiview_registration")
    g = golden("registration")
    vol = g["volume"]
    a = mv.warp_array_to_reference_with_affine_and_sofima_flow_gpu(
        vol, transform_zyx_um=g["transform"], spacing_zyx_um=tuple(g["spacing"]), reference_shape=vol.shape,
        sofima_flow_field_xyz_px=g["flow"], flow_field_stride_zyx_px=g["flow_stride"],
        flow_field_box_start_xyz_px=g["flow_box_start"])
    np.testing.assert_allclose(a, g["flow_warp_constant"], rtol=1e-5, atol=2e-3)
    b = mv.warp_array_to_reference_with_affine_and_sofima_flow_gpu(
        vol, transform_zyx_um=np.eye(4, dtype=np.float32), spacing_zyx_um=(1.0, 1.0, 1.0), reference_shape=vol.shape,
        sofima_flow_field_xyz_px=g["flow"], flow_field_stride_zyx_px=g["flow_stride"],
        flow_field_box_start_xyz_px=g["flow_box_start"], mode="nearest")
    np.testing.assert_allclose(b, g["flow_warp_nearest"], rtol=1e-5, atol=2e-3)
    assert np.mean(b == g["flow_warp_nearest"]) > 0.95
    with pytest.raises(ValueError):
        mv.warp_array_to_reference_with_affine_and_sofima_flow_gpu(
            vol, transform_zyx_um=np.eye(4), spacing_zyx_um=(1, 1, 1), reference_shape=vol.shape,
            sofima_flow_field_xyz_px=np.zeros((2, 2, 2, 2), np.float32), flow_field_stride_zyx_px=(1, 1, 1),
            flow_field_box_start_xyz_px=(0, 0, 0))


def test_decode_warp_dispatch(golden):
    dw = pkg("utils.decode_warping")
    g = golden("registration")
    np.testing.assert_allclose(
        dw.compose_decode_warp_transform_zyx_um(round_transform_zyx_um=g["transform"],
                                                chromatic_transform_zyx_um=g["chromatic"]), g["composed"], atol=1e-6)
    vol = g["volume"].astype(np.uint16)
    out = dw.warp_image_to_reference_frame(vol, transform_zyx_um=np.eye(4, dtype=np.float32),
                                           spacing_zyx_um=(1, 1, 1))
    assert out.dtype == np.float32 and np.array_equal(out, vol.astype(np.float32))
    flow = (g["flow"], {"sofima_status": "identity_fallback_no_error_improvement",
                        "map_stride_zyx_px": list(g["flow_stride"]), "map_box_start_xyz_px": list(g["flow_box_start"]),
                        "reference_shape_zyx_px": list(vol.shape)})
    aff_only = dw.warp_image_to_reference_frame(g["volume"], transform_zyx_um=g["transform"],
                                                spacing_zyx_um=tuple(g["spacing"]), loaded_flow_field=flow)
    np.testing.assert_allclose(aff_only, g["warped_f32"], rtol=1e-6, atol=1e-4)


def test_max_z_projection():
    kernels = pkg("kernels")
    rng = np.random.default_rng(1)
    v = rng.integers(0, 60000, size=(7, 33, 41)).astype(np.uint16)
    got = kernels.max_z_projection(torch.from_numpy(v).to(DEV)).cpu().numpy()
    assert np.array_equal(got, v.max(axis=0).astype(np.float32))


def test_pooled_order_statistics_match_numpy():
    kernels = pkg("kernels")
    rng = np.random.default_rng(2)
    a = rng.gamma(2.0, 50.0, size=70001).astype(np.float32) - 20.0
    b = rng.normal(0, 1, size=1234).astype(np.float32)
    ta, tb = torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)
    pool = kernels.PooledValues(group=False)
    pool.add(ta)
    pool.add(tb)
    allv = np.concatenate([a, b])
    assert pool.count() == allv.size
    assert pool.median() == float(np.median(allv))
    for q in (10.0, 90.0, 0.0, 100.0, 33.3):
        assert abs(pool.percentile(q) - float(np.percentile(allv.astype(np.float64), q))) < 1e-9
    srt = np.sort(allv)
    assert pool.kth([0, 1, 500, allv.size - 1]) == [float(srt[0]), float(srt[1]), float(srt[500]), float(srt[-1])]
    cut = float(np.percentile(a, 10))
    tail = kernels.PooledValues(group=False)
    tail.add(ta, below=cut)
    assert tail.count() == int((a < cut).sum()) and tail.median() == float(np.median(a[a < cut]))
    sub = kernels.PooledValues(group=False)
    sub.add(ta, sub=5.0, above=100.0)
    ref = np.maximum(a - np.float32(5.0), 0)
    assert sub.count() == int((ref > 100.0).sum()) and sub.median() == float(np.median(ref[ref > 100.0]))


def test_global_normalization_matches_reference_run(golden):
    PD = pkg("PixelDecoder")
    g = golden("global_normalization")
    readouts, predictors = g["readouts"], g["predictors"]
    n_bits = readouts.shape[1]
    bits = np.zeros((20, n_bits), dtype=np.int64)
    for c in range(20):
        bits[c, [(c + k * 3) % n_bits for k in range(4)]] = 1
    store = store_from_stack(list(readouts), codebook_frame(bits), predictors=list(predictors))
    dec = PD.PixelDecoder(store, merfish_bits=n_bits, verbose=0)
    dec._global_normalization_vectors(tile_indices=[0, 1, 2], lowpass_sigma=(3.0, 1.0, 1.0))
    norm, bkd = store.load_decode_normalization_vectors(None, "global")
    np.testing.assert_allclose(bkd, g["background"], rtol=2e-5)
    np.testing.assert_allclose(norm, g["normalization"], rtol=2e-5)
    assert dec._global_normalization_loaded
