"""CPU: the oracle restatement against the golden vectors produced by the reference itself
(tests/golden/make_golden.py).  Bit-exact where the reference ran with NumPy arithmetic."""

import warnings

import numpy as np
import pytest

from oracle import decode_ref, philox, rlgc_ref


@pytest.mark.parametrize("name", ["rlgc_a", "rlgc_b"])
def test_rlgc_fullframe_matches_reference_bitwise(golden, name):
    g = golden(name)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out, info = rlgc_ref.rlgc(g["image"], g["psf"], return_info=True)
    assert info.padded_shape == tuple(g["padded_shape"])
    assert np.array_equal(np.asarray(info.pad_width), g["pad_width"])
    assert np.array_equal(out, g["output"])
    assert np.array_equal(rlgc_ref.pad_psf(g["psf"], info.padded_shape), g["padded_psf"])
    # flux doubling of the reference update
    assert abs(out.sum() / g["image"].sum() - 2.0) < 0.02


def test_rlgc_tiled_and_2d_routes(golden):
    g = golden("rlgc_tiled")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = rlgc_ref.chunked_rlgc(g["image"], g["psf"], crop_yx=int(g["crop_yx"]),
                                    rng_seed=int(g["rng_seed"]))
        g2 = golden("rlgc_2d")
        out2 = rlgc_ref.chunked_rlgc(g2["image"], g2["psf"])
    assert np.array_equal(out, g["output"])
    assert out2.shape == g2["image"].shape and np.array_equal(out2, g2["output"])


def test_rlgc_geometry_vectors(golden):
    g = golden("rlgc_geometry")
    assert [rlgc_ref.next_gpu_fft_size(int(s)) for s in g["sizes"]] == list(g["next_size"])
    for length, support, before, after in g["pad_cases"]:
        assert rlgc_ref.axis_linear_fft_padding(int(length), int(support)) == (before, after)
    # config-2 headline shape: 64x2048x2048 with a 64x51x51 PSF -> 128x2187x2187
    assert [s + sum(rlgc_ref.axis_linear_fft_padding(s, p)) for s, p in
            zip((64, 2048, 2048), (64, 51, 51))] == [128, 2187, 2187]


def test_symmetric_boundary_composes_to_one_gather():
    rng = np.random.default_rng(0)
    pad = ((3, 4), (5, 2), (1, 7))  # includes pads larger than the observed extent
    core = (2, 4, 3)
    shape = tuple(c + b + a for c, (b, a) in zip(core, pad))
    vol = rng.normal(size=shape).astype(np.float32)
    ref = vol.copy()
    rlgc_ref.enforce_symmetric_boundary(ref, pad)
    idx = [rlgc_ref.symmetric_source_index(n, b, a) for n, (b, a) in zip(shape, pad)]
    gathered = vol[np.ix_(*idx)]
    assert np.array_equal(ref, gathered)


@pytest.mark.parametrize("name,is3d", [("decode_pixels_16", True), ("decode_pixels_22", True),
                                       ("decode_pixels_2d_none", False)])
def test_decode_pixels_matches_reference(golden, name, is3d):
    g = golden(name)
    assert np.array_equal(decode_ref.lowpass(g["stack"], (3.0, 1.0, 1.0), is3d), g["lowpass"])
    kw = dict(magnitude_threshold=(0.7, 10.0))
    if "background" in g.files:
        kw = dict(background=g["background"], normalization=g["normalization"],
                  excluded_indices=g["excluded_indices"],
                  magnitude_threshold=tuple(g["magnitude_threshold"]))
    dec, mag, dist, sc = decode_ref.decode_pixels(
        g["lowpass"], g["decoding_matrix"], float(g["pixel_threshold"]), arith="standin", **kw)
    assert np.array_equal(dec, g["decoded"])
    assert np.array_equal(mag, g["magnitude"])
    assert np.array_equal(dist, g["distance"])
    assert np.array_equal(sc, g["scaled"])
    # the device arithmetic (fmaf chain + CuPy's double rounding) changes no integer call
    dec2, mag2, dist2, sc2 = decode_ref.decode_pixels(
        g["lowpass"], g["decoding_matrix"], float(g["pixel_threshold"]), arith="device", **kw)
    assert np.array_equal(dec2, g["decoded"])
    for a, b in ((mag2, g["magnitude"]), (dist2, g["distance"]), (sc2, g["scaled"])):
        a32, b32 = a.astype(np.float32), b.astype(np.float32)
        assert np.max(np.abs(a32 - b32)) <= 1e-3  # at most one float16 ulp apart
        assert np.mean(a != b) < 1e-3
    # the float32 distance / magnitude the reference computed before rounding and float16 storage
    # (PixelDecoder.py:2505-2513): the device-arithmetic oracle - which the HIP kernel equals bit for bit - is
    # within north_star's 1e-4 of them
    _d, _m, _di, _s, mag32, dist32 = decode_ref.decode_pixels(
        g["lowpass"], g["decoding_matrix"], float(g["pixel_threshold"]), arith="device", return_f32=True, **kw)
    assert np.max(np.abs(dist32 - g["distance_f32"])) <= 1e-4
    assert np.max(np.abs(mag32 - g["magnitude_f32"])) <= 1e-4 * max(1.0, float(g["magnitude_f32"].max()))


def test_codebook_thresholds_and_normalisation(golden):
    g = golden("decode_pixels_22")
    matrix, genes, pix, tr, blank = decode_ref.load_codebook(g["codebook_bits"], list(g["gene_ids"]))
    assert pix == float(g["pixel_threshold"]) and tr == float(g["transcript_threshold"])
    assert abs(pix - 0.765367) < 1e-6 and abs(tr - 0.605811) < 1e-6
    assert blank == int(g["blank_count"])
    assert np.array_equal(decode_ref.normalize_codebook(matrix), g["decoding_matrix"])


def test_warp_pixel_known_answer(golden):
    g = golden("warp_pixel")
    got = decode_ref.warp_pixel(g["pixel"], g["spacing"], g["origin"], g["affine"], g["camera_to_stage"])
    np.testing.assert_allclose(got, g["expected"], rtol=0, atol=1e-5)
    got = decode_ref.warp_pixel(g["pixel"], g["spacing"], g["origin"], g["affine"], None)
    np.testing.assert_allclose(got, g["expected_no_camera"], rtol=0, atol=1e-5)


def test_philox_known_answer_and_binomial_moments():
    # Random123 known-answer vectors for philox4x32-10
    out = philox.philox4x32_10([0], [0], [0], [0], 0, 0)
    assert [int(o[0]) for o in out] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    out = philox.philox4x32_10([0xFFFFFFFF], [0xFFFFFFFF], [0xFFFFFFFF], [0xFFFFFFFF],
                               0xFFFFFFFF, 0xFFFFFFFF)
    assert [int(o[0]) for o in out] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    n = np.full(200000, 300, dtype=np.int64)
    s = philox.binomial_half(n, np.arange(n.size), 3, 42)
    assert abs(s.mean() - 150.0) < 0.1 and abs(s.var() - 75.0) < 1.0
    assert s.min() >= 0 and s.max() <= 300
    edge = philox.binomial_half(np.array([0, 1, 127, 128, 129, 65535]), np.arange(6), 0, 1)
    assert edge[0] == 0 and edge[1] in (0, 1) and 32767 - 1300 < edge[5] < 32767 + 1300   # 10 sigma


@pytest.mark.parametrize("is3d", [True, False])
def test_ccl_oracle_agrees_with_scipy_labelling(is3d):
    """scikit-image / cuCIM are not installed; SciPy's ``ndimage.label`` with a full structuring element is
    the same partition (skimage documents connectivity = ndim as the full neighbourhood), labelled per
    codeword and renumbered in raster order of each region's first voxel."""
    import scipy.ndimage as ndi

    from oracle import ccl_ref

    rng = np.random.default_rng(17)
    dec = rng.integers(-1, 4, size=(6, 17, 19)).astype(np.int16)
    dec[rng.uniform(size=dec.shape) < 0.45] = -1
    labels, n = ccl_ref.label_equal(dec, is_3d=is3d)
    want = np.zeros(dec.shape, dtype=np.int64)
    planes = [slice(None)] if is3d else [slice(z, z + 1) for z in range(dec.shape[0])]
    structure = np.ones((3, 3, 3))
    if not is3d:
        structure[0] = structure[2] = 0      # 8-connectivity inside the plane only
    nxt = 0
    for sl in planes:
        for code in np.unique(dec[sl][dec[sl] >= 0]):
            lab, k = ndi.label(dec[sl] == code, structure=structure)
            want[sl][lab > 0] = lab[lab > 0] + nxt
            nxt += k
    assert n == nxt
    # same partition <=> a bijection between the two label sets over the foreground
    fg = dec >= 0
    pairs = np.unique(np.stack([labels[fg], want[fg]]), axis=1)
    assert pairs.shape[1] == n and np.unique(pairs[0]).size == n and np.unique(pairs[1]).size == n
    assert (labels[~fg] == 0).all()
    # scikit-image numbering: labels increase with the raster position of each region's first voxel
    # (per plane with a running offset in the 2-D mode)
    first = np.full(n + 1, dec.size, dtype=np.int64)
    np.minimum.at(first, labels.ravel(), np.arange(dec.size))
    assert (np.diff(first[1:]) > 0).all()


def test_labels_and_region_table_match_scikit_image_golden(golden):
    """``label`` / ``remove_small_objects`` / ``regionprops_table`` as computed by the real
    scikit-image 0.18.3 (tests/golden/make_golden_skimage.py) on the reference-run decode fixture."""
    from oracle import ccl_ref

    s = golden("skimage_regionprops")
    g = golden("decode_pixels_16")
    lab3, n3 = ccl_ref.label_equal(g["decoded"], is_3d=True)
    np.testing.assert_array_equal(lab3, s["labels3"])
    lab2, n2 = ccl_ref.label_equal(g["decoded"], is_3d=False)
    np.testing.assert_array_equal(lab2, s["labels2"])
    kept = lab3.copy()
    counts = np.bincount(kept.ravel())
    kept[np.isin(kept, np.nonzero(counts <= 2)[0])] = 0
    np.testing.assert_array_equal(kept, s["kept"])
    df = ccl_ref.region_table(kept, np.moveaxis(g["scaled"], 0, -1), g["magnitude"], g["distance"])
    np.testing.assert_array_equal(df["label"].to_numpy(), s["label"])
    np.testing.assert_array_equal(df["area"].to_numpy(), s["area"])
    cen = df[[f"centroid-{k}" for k in range(3)]].to_numpy()
    np.testing.assert_allclose(cen, s["centroid"], rtol=0, atol=1e-12)
    B = g["scaled"].shape[0]
    im = df[[f"intensity_mean-{k}" for k in range(B)]].to_numpy().astype(np.float64)
    np.testing.assert_array_equal(im, s["intensity_mean"])          # float16 means, bit for bit
    ev = df[[f"inertia_tensor_eigvals-{k}" for k in range(3)]].to_numpy()
    np.testing.assert_allclose(ev, s["inertia_tensor_eigvals"], rtol=1e-9, atol=1e-9)
    np.testing.assert_array_equal(df["magnitude_mean"].to_numpy().astype(np.float64), s["magnitude_mean"])
    np.testing.assert_array_equal(df["distance_min"].to_numpy().astype(np.float32), s["distance_min"])


def test_phase_correlation_matches_scikit_image_golden(golden):
    """FFT sign conventions, peak -> shift wrap and the upsampled-DFT refinement against the real
    scikit-image 0.18.3 (its un-normalised correlation; ``normalization="phase"`` and ``disambiguate``
    came later and stay restated from the published source)."""
    from oracle import registration_ref

    s = golden("skimage_phase_correlation")
    for i in range(4):
        ref, mov = s[f"ref_{i}"], s[f"mov_{i}"]
        for up in (1, 10):
            got = registration_ref.phase_cross_correlation(ref, mov, upsample_factor=up, disambiguate=False,
                                                           normalization=None)
            np.testing.assert_allclose(got, s[f"shift_{i}_up{up}"], rtol=0, atol=1e-6)
        # the variant the reference calls agrees on these well-posed cases (one upsampling step at most)
        # (the test images are circular shifts, so the disambiguated wrap is arbitrary: compare modulo the shape)
        full = registration_ref.phase_cross_correlation(ref, mov, upsample_factor=10, disambiguate=True)
        n = np.asarray(ref.shape, dtype=np.float64)
        wrap = lambda d: (d + n / 2) % n - n / 2
        assert np.abs(wrap(full - s[f"shift_{i}_up10"])).max() <= 0.1 + 1e-5
        assert np.abs(wrap(full + s[f"true_{i}"])).max() <= 0.2 + 1e-5


def test_affine_plus_flow_warp_restatement_matches_the_reference_run(golden):
    """tests/oracle_pipeline.py::warp_with_affine_and_flow (multiview_registration.py:905-1131) against the reference
    run behind registration.npz: constant boundary with a real affine and anisotropic spacing, and the estimator's
    identity / unit-spacing / nearest form."""
    import oracle_pipeline as op

    g = golden("registration")
    a = op.warp_with_affine_and_flow(g["volume"], g["transform"], tuple(g["spacing"]), g["volume"].shape, g["flow"],
                                     g["flow_stride"], g["flow_box_start"])
    np.testing.assert_allclose(a, g["flow_warp_constant"], rtol=1e-5, atol=2e-3)
    b = op.warp_with_affine_and_flow(g["volume"], np.eye(4, dtype=np.float32), (1.0, 1.0, 1.0), g["volume"].shape,
                                     g["flow"], g["flow_stride"], g["flow_box_start"], mode="nearest")
    np.testing.assert_allclose(b, g["flow_warp_nearest"], rtol=1e-5, atol=2e-3)
