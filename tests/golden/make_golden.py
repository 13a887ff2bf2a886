#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference itself.

Run ONLY in the authoring container (``/root/reference`` is absent on the GPU
box; the fixtures are what travels):

    python tests/golden/make_golden.py

The reference's hot path imports CuPy / cuCIM / cuVS at module top, none of
which exist here.  Following the reference's own test technique
(``tests/test_illumination_correction.py:51``,
``tests/test_optimization_codeword_exclusions.py:142-145``) the script installs
stand-in modules in ``sys.modules`` - ``cupy`` forwards to NumPy,
``cupyx.scipy.ndimage`` to SciPy, ``cuvs...pairwise_distance`` to an expanded
float32 distance - and then imports and *executes the reference's own code*.
What the fixtures pin is therefore the reference's control flow and semantics
with NumPy/SciPy arithmetic: integer outputs exactly, floats to ~1e-5
(cuRAND streams, cuFFT rounding and cuVS summation order are not reproducible
off NVIDIA hardware; see DESIGN.md "Oracle pinning").

No reference source is copied: only inputs and the outputs it produced are
saved.
"""

from __future__ import annotations

import sys
import types
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pandas as pd
import scipy.ndimage as sndi

HERE = Path(__file__).resolve().parent
REF_SRC = Path("/root/reference/src")


# ----------------------------------------------------------------------------
# stand-in modules
# ----------------------------------------------------------------------------
class _NullCtx:
    id = 0

    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def use(self):
        pass

    def synchronize(self):
        pass


class _Pool:
    def free_all_blocks(self):
        pass


def _expanded_euclidean(x, y, metric="euclidean", out=None):
    """cuVS 'euclidean' = expanded sqrt(max(0, |x|^2 + |y|^2 - 2 x.y)) in float32."""
    x = np.asarray(x, dtype=np.float32)
    y = np.asarray(y, dtype=np.float32)
    xn = np.sum(x * x, axis=1, dtype=np.float32)[:, None]
    yn = np.sum(y * y, axis=1, dtype=np.float32)[None, :]
    d2 = xn + yn - np.float32(2.0) * (x @ y.T)
    d = np.sqrt(np.maximum(d2, np.float32(0.0))).astype(np.float32)
    if out is not None:
        out[...] = d
        return out
    return d


def install_standins() -> None:
    cp = types.ModuleType("cupy")

    def _getattr(name):
        return getattr(np, name)

    cp.__getattr__ = _getattr  # type: ignore[attr-defined]
    cp.asnumpy = np.asarray
    cp.get_array_module = lambda *a: np
    cp.ElementwiseKernel = lambda *a, **k: (
        lambda recon, ht, cons: np.where(cons < 0, recon, recon * ht).astype(np.float32)
    )
    cp.cuda = SimpleNamespace(
        Device=_NullCtx,
        Stream=SimpleNamespace(null=_NullCtx()),
        memory=SimpleNamespace(OutOfMemoryError=type("OutOfMemoryError", (Exception,), {})),
        runtime=SimpleNamespace(getDeviceCount=lambda: 1),
    )
    cp.get_default_memory_pool = lambda: _Pool()
    cp.get_default_pinned_memory_pool = lambda: _Pool()
    fft = types.ModuleType("cupy.fft")
    fft.rfftn = np.fft.rfftn
    fft.irfftn = np.fft.irfftn
    fft.fftn = np.fft.fftn
    fft.ifftn = np.fft.ifftn
    fft.ifftshift = np.fft.ifftshift
    fft.fftshift = np.fft.fftshift
    fft.config = SimpleNamespace(
        get_plan_cache=lambda: SimpleNamespace(clear=lambda: None)
    )
    cp.fft = fft
    cp.random = np.random
    cp.linalg = np.linalg
    cp.ndarray = np.ndarray
    sys.modules["cupy"] = cp
    sys.modules["cupy.fft"] = fft

    cupyx = types.ModuleType("cupyx")
    cupyx_scipy = types.ModuleType("cupyx.scipy")
    cupyx_ndi = types.ModuleType("cupyx.scipy.ndimage")
    for name in ("gaussian_filter", "affine_transform", "map_coordinates"):
        setattr(cupyx_ndi, name, getattr(sndi, name))
    cupyx_scipy.ndimage = cupyx_ndi
    cupyx.scipy = cupyx_scipy
    sys.modules["cupyx"] = cupyx
    sys.modules["cupyx.scipy"] = cupyx_scipy
    sys.modules["cupyx.scipy.ndimage"] = cupyx_ndi

    def _mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    def _absent(*a, **k):
        raise RuntimeError("third-party routine not available in this container")

    _mod("cucim")
    _mod("cucim.skimage")
    _mod("cucim.skimage.measure", label=_absent, regionprops_table=_absent)
    _mod("cucim.skimage.morphology", remove_small_objects=_absent)
    _mod("cucim.skimage.registration", phase_cross_correlation=_absent)
    _mod("cuvs")
    _mod("cuvs.distance", pairwise_distance=_expanded_euclidean)
    _mod("roifile", roiread=_absent)
    _mod("rtree")
    _mod("shapely")
    _mod("shapely.geometry", Point=object, Polygon=object)
    _mod("skimage")
    _mod("skimage.draw", polygon=_absent)
    _mod("skimage.measure", regionprops_table=_absent)

    # jax: the reference's SOFIMA glue uses jax.numpy as an array library only (no autodiff);
    # NumPy + scipy.ndimage.map_coordinates have the same semantics for the calls it makes
    jnp = _mod("jax.numpy")
    jnp.__getattr__ = lambda name: getattr(np, name)  # type: ignore[attr-defined]
    jsp_ndi = _mod("jax.scipy.ndimage", map_coordinates=sndi.map_coordinates)
    jsp = _mod("jax.scipy", ndimage=jsp_ndi)
    _mod("jax", numpy=jnp, scipy=jsp, jit=lambda f: f, device_get=lambda x: x)


def import_reference():
    install_standins()
    sys.path.insert(0, str(REF_SRC))
    import importlib

    rl = importlib.import_module("merfish3danalysis.utils.rlgc")
    mv = importlib.import_module("merfish3danalysis.utils.multiview_registration")
    dw = importlib.import_module("merfish3danalysis.utils.decode_warping")
    sf = importlib.import_module("merfish3danalysis.utils.sofima_registration")
    pdm = importlib.import_module("merfish3danalysis.PixelDecoder")
    return rl, mv, dw, sf, pdm


# ----------------------------------------------------------------------------
# synthetic inputs (shared with the tests through the saved arrays)
# ----------------------------------------------------------------------------
def gaussian_psf(shape, sigma):
    zz, yy, xx = np.meshgrid(
        *[np.arange(s, dtype=np.float64) - (s // 2) for s in shape], indexing="ij"
    )
    psf = np.exp(
        -0.5 * ((zz / sigma[0]) ** 2 + (yy / sigma[1]) ** 2 + (xx / sigma[2]) ** 2)
    )
    return (psf / psf.sum()).astype(np.float32)


def phantom(shape, n_spots, seed, psf, background=20.0, amp=(400.0, 1500.0)):
    rng = np.random.default_rng(seed)
    vol = np.zeros(shape, dtype=np.float64)
    for _ in range(n_spots):
        z, y, x = (rng.integers(0, s) for s in shape)
        vol[z, y, x] += rng.uniform(*amp)
    blurred = sndi.convolve(vol, psf.astype(np.float64), mode="reflect") * psf.size ** 0
    noisy = rng.poisson(blurred * 40.0 + background)
    return np.clip(noisy, 0, 65535).astype(np.uint16)


def hw4_codebook(n_bits: int, n_codes: int, seed: int, n_blank: int = 10):
    """Greedy weight-4, Hamming-distance-4 codebook as a DataFrame (gene_id + bits)."""
    import itertools

    rng = np.random.default_rng(seed)
    combos = list(itertools.combinations(range(n_bits), 4))
    rng.shuffle(combos)
    chosen = []
    for c in combos:
        s = set(c)
        if all(len(s & set(o)) <= 2 for o in chosen):
            chosen.append(c)
        if len(chosen) == n_codes:
            break
    mat = np.zeros((len(chosen), n_bits), dtype=np.int64)
    for i, c in enumerate(chosen):
        mat[i, list(c)] = 1
    names = [f"gene{i:03d}" for i in range(len(chosen) - n_blank)] + [
        f"Blank-{i:02d}" for i in range(n_blank)
    ]
    df = pd.DataFrame(mat, columns=[f"bit{i + 1:02d}" for i in range(n_bits)])
    df.insert(0, "gene_id", names)
    return df


# ----------------------------------------------------------------------------
# fixtures
# ----------------------------------------------------------------------------
def make_rlgc(rl):
    cases = {
        "rlgc_a": dict(shape=(12, 40, 44), psf_shape=(7, 9, 9), sigma=(1.4, 1.6, 1.6), seed=3, n=25),
        "rlgc_b": dict(shape=(16, 64, 64), psf_shape=(9, 11, 11), sigma=(1.8, 1.6, 1.6), seed=4, n=60),
    }
    for name, c in cases.items():
        psf = gaussian_psf(c["psf_shape"], c["sigma"])
        img = phantom(c["shape"], c["n"], c["seed"], psf)
        out = rl.chunked_rlgc(img, psf, crop_yx=2048, rng_seed=42, release_memory=False)
        padded, pad_width = rl.pad_for_linear_fft(img, psf.shape)
        np.savez_compressed(
            HERE / f"{name}.npz",
            image=img,
            psf=psf,
            output=out.astype(np.float32),
            pad_width=np.asarray(pad_width, dtype=np.int64),
            padded_shape=np.asarray(padded.shape, dtype=np.int64),
            padded_psf=rl.pad_psf(psf, padded.shape).astype(np.float32),
        )
        print(name, out.shape, float(out.sum()) / float(img.sum()))
    # tiled route: crop smaller than the image -> per-tile seeds and halos
    psf = gaussian_psf((5, 7, 7), (1.2, 1.3, 1.3))
    img = phantom((8, 40, 56), 30, 5, psf)
    out = rl.chunked_rlgc(img, psf, crop_yx=24, rng_seed=7, release_memory=False)
    np.savez_compressed(
        HERE / "rlgc_tiled.npz", image=img, psf=psf, output=out.astype(np.float32),
        crop_yx=np.int64(24), rng_seed=np.int64(7),
    )
    # 2-D input promoted to one z plane
    psf2 = gaussian_psf((1, 9, 9), (1.0, 1.5, 1.5))[0]
    img2 = phantom((1, 48, 48), 12, 6, psf2[None])[0]
    out2 = rl.chunked_rlgc(img2, psf2, rng_seed=42, release_memory=False)
    np.savez_compressed(HERE / "rlgc_2d.npz", image=img2, psf=psf2, output=out2.astype(np.float32))
    # helper vectors
    sizes = np.arange(1, 2200, dtype=np.int64)
    np.savez_compressed(
        HERE / "rlgc_geometry.npz",
        sizes=sizes,
        next_size=np.asarray([rl.next_gpu_fft_size(int(s)) for s in sizes]),
        pad_cases=np.asarray(
            [
                (l, p, *rl._axis_linear_fft_padding(l, p))
                for l in (1, 16, 64, 100, 256, 2048)
                for p in (1, 7, 16, 51, 64)
            ],
            dtype=np.int64,
        ),
    )


def _bare_decoder(pdm, codebook_df, n_bits, is_3d=True):
    dec = pdm.PixelDecoder.__new__(pdm.PixelDecoder)
    dec._datastore = SimpleNamespace(codebook=codebook_df)
    dec._n_merfish_bits = n_bits
    dec._verbose = 0
    dec._is_3D = is_3d
    dec._load_codebook()
    dec._excluded_gene_ids, dec._excluded_codeword_indices = dec._resolve_excluded_gene_ids(None)
    dec._decoding_matrix_no_errors = dec._normalize_codebook(include_errors=False)
    dec._decoding_matrix = dec._decoding_matrix_no_errors.copy()
    dec._iterative_normalization_loaded = False
    dec._global_normalization_loaded = False
    return dec


def synth_bit_stack(codebook_df, n_bits, shape, n_spots, seed):
    """f32[B,Z,Y,X] 'readout x predictor' stack with coded spots + background."""
    rng = np.random.default_rng(seed)
    mat = codebook_df.iloc[:, 1 : n_bits + 1].to_numpy(dtype=np.int64)
    stack = rng.gamma(2.0, 15.0, size=(n_bits, *shape))
    truth = []
    for _ in range(n_spots):
        code = int(rng.integers(0, mat.shape[0]))
        z, y, x = (int(rng.integers(1, s - 1)) for s in shape)
        amp = rng.uniform(600, 2500)
        blob = np.zeros(shape)
        blob[z, y, x] = amp
        blob = sndi.gaussian_filter(blob, (0.9, 1.3, 1.3)) * 12.0
        for b in np.nonzero(mat[code])[0]:
            stack[b] += blob * rng.uniform(0.7, 1.3)
        truth.append((code, z, y, x))
    return stack.astype(np.float32), np.asarray(truth, dtype=np.int64)


def _record_f32(dec):
    """Keep what ``_calculate_distances`` / ``_normalize_pixel_traces`` return per plane: the reference rounds to 5
    decimals and stores float16 (PixelDecoder.py:2621-2632), the float32 values are what "distances within 1e-4"
    is about."""
    rec = {"distance": [], "magnitude": []}
    calc, norm = dec._calculate_distances, dec._normalize_pixel_traces

    def calc_rec(*a, **k):
        d, i = calc(*a, **k)
        rec["distance"].append(np.asarray(d, dtype=np.float32).copy())
        return d, i

    def norm_rec(*a, **k):
        n, m = norm(*a, **k)
        rec["magnitude"].append(np.asarray(m, dtype=np.float32).copy())
        return n, m

    dec._calculate_distances, dec._normalize_pixel_traces = calc_rec, norm_rec
    return rec


def make_decode(pdm):
    for name, n_bits, n_codes, shape, seed, excl in (
        ("decode_pixels_16", 16, 140, (6, 48, 52), 11, None),
        ("decode_pixels_22", 22, 140, (4, 40, 64), 12, ("gene003", "gene010")),
    ):
        cb = hw4_codebook(n_bits, n_codes, seed)
        dec = _bare_decoder(pdm, cb, n_bits)
        if excl:
            dec._excluded_gene_ids, dec._excluded_codeword_indices = dec._resolve_excluded_gene_ids(excl)
        stack, truth = synth_bit_stack(cb, n_bits, shape, 40, seed)
        # reference low-pass under the SciPy stand-in
        dec._image_data = stack
        dec._lp_filter(sigma=(3.0, 1.0, 1.0))
        lp = dec._image_data_lp.copy()
        rng = np.random.default_rng(seed + 100)
        bkd = rng.uniform(5, 40, size=n_bits).astype(np.float32)
        nrm = rng.uniform(150, 400, size=n_bits).astype(np.float32)
        dec._global_background_vector = bkd
        dec._global_normalization_vector = nrm
        dec._global_normalization_loaded = True
        mag_thr = (0.9, 10.0)
        f32 = _record_f32(dec)
        dec._decode_pixels(magnitude_threshold=mag_thr)
        np.savez_compressed(
            HERE / f"{name}.npz",
            distance_f32=np.stack(f32["distance"]).reshape(dec._distance_image.shape),
            magnitude_f32=np.stack(f32["magnitude"]).reshape(dec._magnitude_image.shape),
            codebook_bits=cb.iloc[:, 1 : n_bits + 1].to_numpy(dtype=np.int8),
            gene_ids=np.asarray(cb["gene_id"].tolist()),
            stack=stack,
            truth=truth,
            lowpass=lp.astype(np.float32),
            background=bkd,
            normalization=nrm,
            magnitude_threshold=np.asarray(mag_thr, dtype=np.float64),
            pixel_threshold=np.float64(dec._pixel_assignment_threshold),
            transcript_threshold=np.float64(dec._transcript_distance_threshold),
            excluded_indices=np.asarray(dec._excluded_codeword_indices, dtype=np.int64),
            decoding_matrix=np.asarray(dec._decoding_matrix, dtype=np.float64),
            decoded=dec._decoded_image,
            magnitude=dec._magnitude_image,
            distance=dec._distance_image,
            scaled=dec._scaled_pixel_images,
            blank_count=np.int64(dec._blank_count),
        )
        print(name, "decoded px:", int((dec._decoded_image >= 0).sum()))
    # un-normalised decode ("none") on a small stack, 2-D low-pass mode
    cb = hw4_codebook(16, 60, 21)
    dec = _bare_decoder(pdm, cb, 16, is_3d=False)
    stack, _ = synth_bit_stack(cb, 16, (3, 32, 36), 12, 21)
    stack /= np.float32(300.0)
    dec._image_data = stack
    dec._lp_filter(sigma=(3.0, 1.0, 1.0))
    lp2 = dec._image_data_lp.copy()
    f32 = _record_f32(dec)
    dec._decode_pixels(magnitude_threshold=(0.7, 10.0))
    np.savez_compressed(
        HERE / "decode_pixels_2d_none.npz",
        distance_f32=np.stack(f32["distance"]).reshape(dec._distance_image.shape),
        magnitude_f32=np.stack(f32["magnitude"]).reshape(dec._magnitude_image.shape),
        codebook_bits=cb.iloc[:, 1:17].to_numpy(dtype=np.int8),
        stack=stack,
        lowpass=lp2,
        decoded=dec._decoded_image,
        magnitude=dec._magnitude_image,
        distance=dec._distance_image,
        scaled=dec._scaled_pixel_images,
        pixel_threshold=np.float64(dec._pixel_assignment_threshold),
        decoding_matrix=np.asarray(dec._decoding_matrix, dtype=np.float64),
    )
    # _warp_pixel known-answer (tests/test_pixeldecoder_coordinates.py:6-41)
    pixel = np.array([10.0, 20.0, 30.0], dtype=np.float32)
    spacing = np.array([0.32, 0.098, 0.098], dtype=np.float32)
    origin = np.array([2761.3, 107.81, 0.0], dtype=np.float32)
    c2s = np.array([[1, 0, 0, 0], [0, -0.07, -1, 0], [0, -1, 0.07, 0], [0, 0, 0, 1]], dtype=np.float32)
    ga = np.array([[1, 0, 0, 0.5], [0, 1, 0, 1.25], [0, 0, 1, -2.0], [0, 0, 0, 1]], dtype=np.float32)
    np.savez_compressed(
        HERE / "warp_pixel.npz", pixel=pixel, spacing=spacing, origin=origin,
        camera_to_stage=c2s, affine=ga,
        expected=pdm.PixelDecoder._warp_pixel(pixel, spacing, origin, ga, c2s),
        expected_no_camera=pdm.PixelDecoder._warp_pixel(pixel, spacing, origin, ga, None),
    )


# ----------------------------------------------------------------------------
# _extract_barcodes: the reference's own table assembly over pinned library arithmetic
# ----------------------------------------------------------------------------
def _skimage_standins():
    """``label`` / ``remove_small_objects`` / ``regionprops_table`` with the scikit-image API the reference calls,
    computed by ``oracle.ccl_ref`` - whose labels, areas, centroids, float16 ``intensity_mean``, eigenvalues and
    ``intensity_min`` are themselves pinned bit for bit to the real scikit-image 0.18.3
    (tests/golden/skimage_regionprops.npz).  Only the arithmetic the reference delegates to cuCIM / scikit-image is
    stood in; everything from PixelDecoder.py:3062 on is the reference's own code."""
    sys.path.insert(0, str(HERE.parents[1]))
    from oracle import ccl_ref

    def label(image, background=-1, connectivity=None, return_num=False):
        img = np.asarray(image)
        assert background == -1
        if img.ndim == 3:
            assert connectivity == 3
            lab, n = ccl_ref.label_equal(img, True)
        else:
            assert connectivity == 2
            lab, n = ccl_ref.label_equal(img[None], False)
            lab = lab[0]
        return (lab, n) if return_num else lab

    def remove_small_objects(labels, max_size=0):
        out = np.array(labels, copy=True)
        counts = np.bincount(out.ravel())
        small = np.nonzero(counts <= int(max_size))[0]
        small = small[small != 0]
        if small.size:
            out[np.isin(out, small)] = 0
        return out

    def regionprops_table(label_image, intensity_image=None, properties=()):
        labels = np.asarray(label_image)
        inten = np.asarray(intensity_image)
        multi = inten.ndim == labels.ndim + 1
        chan = inten if multi else inten[..., None]
        zero = np.zeros(labels.shape, dtype=np.float32)
        # region_table computes every property; the distance / magnitude slots get the image when it is the one asked for
        df = ccl_ref.region_table(labels, chan, inten if not multi else zero, inten if not multi else zero)
        out = {}
        for prop in properties:
            if prop == "label":
                out["label"] = df["label"].to_numpy(dtype=np.int64) if len(df) else np.zeros(0, np.int64)
            elif prop == "area":
                out["area"] = df["area"].to_numpy(dtype=np.float64) if len(df) else np.zeros(0)
            elif prop == "centroid":
                for k in range(labels.ndim):
                    out[f"centroid-{k}"] = df[f"centroid-{k}"].to_numpy() if len(df) else np.zeros(0)
            elif prop == "intensity_mean":
                if multi:
                    for k in range(chan.shape[-1]):
                        out[f"intensity_mean-{k}"] = (df[f"intensity_mean-{k}"].to_numpy() if len(df)
                                                      else np.zeros(0, chan.dtype))
                else:
                    out["intensity_mean"] = df["magnitude_mean"].to_numpy() if len(df) else np.zeros(0, inten.dtype)
            elif prop == "intensity_min":
                out["intensity_min"] = df["distance_min"].to_numpy() if len(df) else np.zeros(0, np.float32)
            elif prop == "inertia_tensor_eigvals":
                for k in range(3):
                    out[f"inertia_tensor_eigvals-{k}"] = (df[f"inertia_tensor_eigvals-{k}"].to_numpy() if len(df)
                                                          else np.zeros(0))
            else:
                raise KeyError(prop)
        return out

    return label, remove_small_objects, regionprops_table


def make_extract(pdm):
    """Reference ``_extract_barcodes`` (PixelDecoder.py:2908-3201) on the volumes of the decode fixtures ->
    tests/golden/extract_barcodes_{16,22,2d}.npz (one array per table column)."""
    label, rso, rpt = _skimage_standins()
    pdm.label, pdm.remove_small_objects = label, rso
    pdm.gpu_regionprops_table, pdm.regionprops_table = rpt, rpt
    spacing = np.array([0.315, 0.098, 0.098], dtype=np.float32)
    origin = np.array([2761.3, 107.81, 12.5], dtype=np.float32)
    c2s = np.array([[1, 0, 0, 0], [0, -0.07, -1, 0], [0, -1, 0.07, 0], [0, 0, 0, 1]], dtype=np.float32)
    ga = np.array([[1, 0, 0, 0.5], [0, 1, 0, 1.25], [0, 0, 1, -2.0], [0, 0, 0, 1]], dtype=np.float32)
    for name, src, n_bits, is_3d, min_px, tile_idx in (
        ("extract_barcodes_16", "decode_pixels_16", 16, True, 3, 2),
        ("extract_barcodes_22", "decode_pixels_22", 22, True, 4, 0),
        ("extract_barcodes_2d", "decode_pixels_16", 16, False, 2, 5),
    ):
        d = np.load(HERE / f"{src}.npz")
        cb = pd.DataFrame(d["codebook_bits"].astype(np.int64), columns=[f"bit{i + 1:02d}" for i in range(n_bits)])
        cb.insert(0, "gene_id", [str(g) for g in d["gene_ids"]])
        dec = _bare_decoder(pdm, cb, n_bits, is_3d=is_3d)
        dec._decoded_image = d["decoded"]
        dec._scaled_pixel_images = d["scaled"]
        dec._distance_image = d["distance"]
        dec._magnitude_image = d["magnitude"]
        dec._optimize_normalization_weights = False
        dec._collect_chromatic_centroids = False
        dec._tile_idx = tile_idx
        dec._z_crop = False
        dec._spacing, dec._origin, dec._affine, dec._camera_to_stage_affine = spacing, origin, ga, c2s
        dec._extract_barcodes(minimum_pixels=min_px, maximum_pixels=500)
        df = dec._df_barcodes
        out = {"columns": np.asarray(list(df.columns)), "minimum_pixels": np.int64(min_px), "tile_idx": np.int64(tile_idx),
               "is_3d": np.bool_(is_3d), "spacing": spacing, "origin": origin, "affine": ga, "camera_to_stage": c2s,
               "source": np.asarray(src)}
        for c in df.columns:
            v = df[c].to_numpy()
            out["col_" + c] = v.astype(str) if v.dtype == object else v
        np.savez_compressed(HERE / f"{name}.npz", **out)
        print(name, len(df), "transcripts;", list(df.columns)[:6], "...")


def make_optimize(pdm):
    """The reference's ``optimize_normalization_by_decoding`` (PixelDecoder.py:4581-4757, worker :312-414) and a
    following ``decode_one_tile(normalization_method="iterative")`` (:4471-4579), run in this process on the
    package's in-memory datastore double: the spawn-per-GPU helpers are bound to plain calls, ``qi2labDataStore(path)``
    to the double, the scikit-image / cuCIM calls to the pinned stand-ins above.  Saved: inputs, the vectors after the
    global stage and after every iteration, the per-iteration transcript counts and the final tile table."""
    import importlib

    label, rso, rpt = _skimage_standins()
    pdm.label, pdm.remove_small_objects = label, rso
    pdm.gpu_regionprops_table, pdm.regionprops_table = rpt, rpt
    ds_mod = importlib.import_module("merfish3d-analysis_amd.datastore")

    n_bits, n_tiles, shape = 16, 2, (8, 64, 64)
    cb = hw4_codebook(n_bits, 60, 71)
    mat = cb.iloc[:, 1 : n_bits + 1].to_numpy(dtype=np.int64)
    rng = np.random.default_rng(71)
    gain = np.linspace(0.6, 1.7, n_bits)          # per-bit brightness: what the normalisation has to find
    readouts = np.zeros((n_tiles, n_bits, *shape), dtype=np.uint16)
    predictors = np.zeros((n_tiles, n_bits, *shape), dtype=np.float32)
    for t in range(n_tiles):
        vols = np.zeros((n_bits, *shape))
        pred = np.zeros((n_bits, *shape))
        for _ in range(140):
            code = int(rng.integers(0, mat.shape[0]))
            z, y, x = (int(rng.integers(1, s - 1)) for s in shape)
            amp = rng.uniform(900, 2600)
            for b in np.nonzero(mat[code])[0]:
                vols[b, z, y, x] += amp * gain[b] * rng.uniform(0.85, 1.15)
                pred[b, z, y, x] += 1.0
        for b in range(n_bits):
            v = sndi.gaussian_filter(vols[b], (0.9, 1.3, 1.3)) * 12.0 + 110.0
            readouts[t, b] = np.clip(rng.poisson(v), 0, 65535).astype(np.uint16)
            pb = np.clip(sndi.gaussian_filter(pred[b], (0.8, 1.1, 1.1)) * 14.0 + 0.03, 0.0, 1.0)
            predictors[t, b] = (np.round(pb * 256.0) / 256.0).astype(np.float32)   # coarse values: small fixture

    store = ds_mod.InMemoryDataStore(num_tiles=n_tiles, num_rounds=1, num_bits=n_bits, codebook=cb,
                                     voxel_size_zyx_um=(0.315, 0.098, 0.098))
    for t in range(n_tiles):
        for b in range(n_bits):
            store.set_corrected_image(readouts[t, b], tile=t, bit=b)
            store.save_local_feature_predictor_image(predictors[t, b], tile=t, bit=b)
        store.set_stage_position_zyx_um(store.tile_ids[t], np.array([10.0, 50.0 * t, 0.0], dtype=np.float32))
    history = []
    save = store.save_decode_normalization_vectors

    def save_rec(decode_run_key, kind, normalization, background, *a, **k):
        history.append((kind, np.asarray(normalization, dtype=np.float32).copy(),
                        np.asarray(background, dtype=np.float32).copy()))
        return save(decode_run_key, kind, normalization, background, *a, **k)

    store.save_decode_normalization_vectors = save_rec

    # process helpers -> plain calls; the worker's datastore re-open -> the double
    counts = []
    pdm._start_gpu_worker_process = lambda target, args, physical_gpu_id: target(*args)
    pdm._join_gpu_workers = lambda procs, what: None
    pdm.preload_cuda_libraries = lambda: None
    pdm.qi2labDataStore = lambda path, validate=False: store
    real_torch = sys.modules.get("torch")
    sys.modules["torch"] = SimpleNamespace(cuda=SimpleNamespace(set_device=lambda i: None))
    try:
        dec = pdm.PixelDecoder(store, merfish_bits=n_bits, num_gpus=1, verbose=0)
        load_all = dec._load_all_barcodes

        def load_all_rec(*a, **k):
            load_all(*a, **k)
            counts.append(len(dec._df_barcodes_loaded))

        dec._load_all_barcodes = load_all_rec
        kw = dict(lowpass_sigma=(3.0, 1.0, 1.0), magnitude_threshold=(0.9, 10.0), minimum_pixels=3)
        dec.optimize_normalization_by_decoding(n_iterations=3, tile_indices=[0, 1], **kw)
        dec2 = pdm.PixelDecoder(store, merfish_bits=n_bits, num_gpus=1, verbose=0)
        dec2.decode_one_tile(tile_idx=1, display_results=False, return_results=False, normalization_method="iterative",
                             **kw)
        table = dec2._df_barcodes
    finally:
        if real_torch is not None:
            sys.modules["torch"] = real_torch
        else:
            sys.modules.pop("torch", None)
    out = {"codebook_bits": mat.astype(np.int8), "gene_ids": np.asarray(cb["gene_id"].tolist()),
           "readouts": readouts, "predictors": predictors, "kinds": np.asarray([h[0] for h in history]),
           "normalization_history": np.stack([h[1] for h in history]),
           "background_history": np.stack([h[2] for h in history]), "barcode_counts": np.asarray(counts, dtype=np.int64),
           "decoded": dec2._decoded_image, "table_columns": np.asarray(list(table.columns))}
    for c in table.columns:
        v = table[c].to_numpy()
        out["col_" + c] = v.astype(str) if v.dtype == object else v
    np.savez_compressed(HERE / "optimize_normalization.npz", **out)
    print("optimize:", [h[0] for h in history], counts, len(table), "transcripts in tile 1")
    print(" final normalization", np.round(history[-1][1], 1))


class _FakeStore:
    """Duck-typed datastore for the normalisation fixtures (identity transforms)."""

    def __init__(self, readouts, predictors, n_bits):
        self._r, self._p = readouts, predictors
        self.tile_ids = [f"tile{i:04d}" for i in range(readouts.shape[0])]
        self.bit_ids = [f"bit{i + 1:03d}" for i in range(n_bits)]
        self.round_ids = ["round001"]
        self.voxel_size_zyx_um = (0.315, 0.098, 0.098)
        self.saved = {}

    def _t(self, tile):
        return self.tile_ids.index(tile) if isinstance(tile, str) else int(tile)

    def _b(self, bit):
        return self.bit_ids.index(bit) if isinstance(bit, str) else int(bit)

    def load_local_readout_image(self, tile, bit, return_future=False):
        return self._r[self._t(tile), self._b(bit)]

    def load_local_feature_predictor_image(self, tile, bit, return_future=False):
        return self._p[self._t(tile), self._b(bit)]

    def load_local_wavelengths_um(self, tile, bit):
        return (0.561, 0.590)

    def load_local_round_linker(self, tile, bit):
        return 1

    def load_chromatic_affine_transform_zyx_um(self, wavelength_um):
        return np.eye(4, dtype=np.float32)

    def load_local_sofima_flow_field(self, tile, round, return_future=False):
        return None

    def save_decode_normalization_vectors(self, key, kind, norm, bkd, **kw):
        self.saved[kind] = (np.asarray(norm).copy(), np.asarray(bkd).copy())


def make_normalization(pdm):
    n_bits, tiles, shape = 16, 3, (6, 40, 40)
    rng = np.random.default_rng(31)
    readouts = rng.gamma(2.0, 60.0, size=(tiles, n_bits, *shape))
    readouts[0, 2, 3, 5, 5] = 60000.0  # hot pixel (> 50000) (PixelDecoder.py:1072-1074)
    readouts = np.clip(readouts, 0, 65535).astype(np.uint16)
    predictors = rng.uniform(0.0, 1.0, size=(tiles, n_bits, *shape)).astype(np.float32)
    store = _FakeStore(readouts, predictors, n_bits)
    cb = hw4_codebook(n_bits, 60, 33)
    dec = _bare_decoder(pdm, cb, n_bits)
    dec._datastore = store
    dec._z_slice = slice(0, None)
    dec._decode_run_key = None
    dec._effective_decode_mode = "3d"
    dec._normalization_cell_polygons = []
    dec._normalization_cell_segmentation_present = False
    dec._normalization_cell_mask_cache = {}
    dec._normalization_cell_mask_for_tile = lambda tile_id, image_shape_zyx: None
    dec._global_normalization_vectors(tile_indices=[0, 1, 2], lowpass_sigma=(3.0, 1.0, 1.0))
    norm, bkd = store.saved["global"]
    np.savez_compressed(
        HERE / "global_normalization.npz", readouts=readouts, predictors=predictors,
        normalization=norm, background=bkd,
    )
    print("global norm", norm[:4], bkd[:4])

    # iterative normalisation from a synthetic barcode table (PixelDecoder.py:1250-1421)
    rng = np.random.default_rng(41)
    mat = cb.iloc[:, 1 : n_bits + 1].to_numpy(dtype=np.int64)
    rows = []
    for i in range(200):
        code = int(rng.integers(0, mat.shape[0]))
        on = np.nonzero(mat[code])[0] + 1
        row = {"gene_id": cb["gene_id"][code]}
        for b in range(1, n_bits + 1):
            base = rng.uniform(0.0, 40.0)
            row[f"bit{b:02d}_mean_intensity"] = float(np.float16(base + (rng.uniform(150, 500) if b in on else 0.0)))
        for j, ob in enumerate(on[:4]):
            row[f"on_bit_{j + 1}"] = int(ob)
        rows.append(row)
    df = pd.DataFrame(rows)
    dec._df_barcodes_loaded = df
    dec._iterative_background_vector = None
    dec._iterative_normalization_vector = None
    dec._global_background_vector = bkd
    dec._global_normalization_vector = norm
    dec._optimization_excluded_gene_ids = ()
    dec._iterative_normalization_vectors()
    inorm, ibkd = store.saved["iterative"]
    df.to_json(HERE / "iterative_normalization_barcodes.json", orient="split")
    np.savez_compressed(
        HERE / "iterative_normalization.npz", normalization=inorm, background=ibkd,
        global_normalization=norm, global_background=bkd,
    )
    print("iter norm", inorm[:4], ibkd[:4])


def make_registration(mv, dw, sf):
    rng = np.random.default_rng(51)
    # overlap slices + point transform (pure NumPy in the reference)
    shapes = [(31, 64, 64), (24, 48, 48), (8, 10, 12)]
    shifts = [(0.0, 3.2, -4.7), (0.0, -8.0, 7.0), (0.0, 20.0, 0.0), (1.5, -0.4, 0.0), (0.0, 0.0, -12.0)]
    rows = []
    for s in shapes:
        for t in shifts:
            sl = mv._overlap_slices_after_translation(s, t)
            rows.append([*s, *t, *([-1] * 6 if sl is None else [v for q in sl for v in (q.start, q.stop)])])
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] += rng.normal(0, 0.01, size=(3, 3)).astype(np.float32)
    T[:3, 3] = (0.63, -0.4, 1.2)
    pts = rng.uniform(0, 40, size=(17, 3)).astype(np.float32)
    spacing = (0.315, 0.098, 0.098)
    # affine warp: u16 in -> u16 out, f32 in -> f32 out (multiview_registration.py:882-892)
    vol = sndi.gaussian_filter(rng.uniform(0, 1, size=(10, 36, 40)), 1.5)
    vol = ((vol - vol.min()) / (vol.max() - vol.min()) * 4000).astype(np.float32)
    warped_f = mv.warp_array_to_reference_gpu(vol, transform_zyx_um=T, spacing_zyx_um=spacing, reference_shape=vol.shape)
    warped_u = mv.warp_array_to_reference_gpu(vol.astype(np.uint16), transform_zyx_um=T, spacing_zyx_um=spacing, reference_shape=(10, 30, 44))
    # flow warp
    flow = rng.normal(0, 1.2, size=(3, 3, 5, 6)).astype(np.float32)
    stride = (4.0, 8.0, 8.0)
    box = (4.0, 4.0, 2.0)
    fw_const = mv.warp_array_to_reference_with_affine_and_sofima_flow_gpu(
        vol, transform_zyx_um=T, spacing_zyx_um=spacing, reference_shape=vol.shape,
        sofima_flow_field_xyz_px=flow, flow_field_stride_zyx_px=stride, flow_field_box_start_xyz_px=box)
    fw_near = mv.warp_array_to_reference_with_affine_and_sofima_flow_gpu(
        vol, transform_zyx_um=np.eye(4, dtype=np.float32), spacing_zyx_um=(1.0, 1.0, 1.0), reference_shape=vol.shape,
        sofima_flow_field_xyz_px=flow, flow_field_stride_zyx_px=stride, flow_field_box_start_xyz_px=box, mode="nearest")
    chrom = np.eye(4, dtype=np.float32)
    chrom[:3, :3] *= np.float32(1.002)
    chrom[:3, 3] = (0.01, -0.03, 0.02)
    np.savez_compressed(
        HERE / "registration.npz",
        overlap_rows=np.asarray(rows, dtype=np.float64),
        transform=T, points=pts, spacing=np.asarray(spacing, dtype=np.float32),
        points_ref=mv.transform_points_to_reference(pts, transform_zyx_um=T, spacing_zyx_um=spacing, origin_zyx_um=(1.0, 2.0, 3.0)),
        volume=vol, warped_f32=warped_f, warped_u16=warped_u,
        flow=flow, flow_stride=np.asarray(stride), flow_box_start=np.asarray(box),
        flow_warp_constant=fw_const, flow_warp_nearest=fw_near,
        chromatic=chrom,
        composed=dw.compose_decode_warp_transform_zyx_um(round_transform_zyx_um=T, chromatic_transform_zyx_um=chrom),
    )
    # SOFIMA glue owned by the reference (pure NumPy parts)
    cfg = sf.SofimaRegistrationConfig()
    ps = [sf._resolve_patch_and_step(s, cfg) for s in [(16, 256, 256), (64, 2048, 2048), (3, 20, 500), (31, 64, 64)]]
    fl = rng.normal(0, 2.0, size=(3, 4, 6, 7)).astype(np.float32)
    fl[2, 1, 2, 3] = 40.0
    fl[2, 0, 0, 0] = np.nan
    stab, meta = sf._stabilize_axial_flow_component(fl, cfg)
    sparse = fl.copy()
    sparse[:, rng.uniform(size=fl.shape[1:]) < 0.4] = np.nan
    a = rng.normal(size=(5, 12, 12)).astype(np.float32)
    b = a + rng.normal(scale=0.1, size=a.shape).astype(np.float32)
    np.savez_compressed(
        HERE / "sofima_glue.npz",
        patch_step=np.asarray(ps, dtype=np.int64),
        flow=fl, stabilized=stab,
        stab_median=np.float64(meta["axial_flow_median_px"]),
        stab_clipped=np.int64(meta["axial_flow_clipped_vectors"]),
        stab_valid=np.int64(meta["axial_flow_valid_vectors"]),
        sparse_flow=sparse, median_init=sf._median_initial_flow_field(sparse),
        nmse_a=a, nmse_b=b, nmse=np.float64(sf._normalized_mean_squared_error(a, b)),
    )


def make_sofima_jax_glue(sf):
    """_compose_flow_fields_same_grid / _refine_flow_vectors_subpixel
    (sofima_registration.py:178-417) run with the NumPy/SciPy stand-in for jax."""
    rng = np.random.default_rng(77)
    cfg = sf.SofimaRegistrationConfig()
    base = rng.normal(0, 2.0, size=(3, 4, 6, 7)).astype(np.float32)
    res = rng.normal(0, 3.0, size=(3, 4, 6, 7)).astype(np.float32)
    stride, box = (5.0, 16.0, 16.0), (16.0, 16.0, 5.0)
    composed = sf._compose_flow_fields_same_grid(base, res, stride_zyx=stride, box_start_xyz=box)
    # sub-pixel refinement: smooth random volume, moving = fixed sampled at a known fractional shift
    shape, patch, step = (16, 40, 44), (6, 12, 12), (3, 6, 6)
    fixed = sndi.gaussian_filter(rng.normal(size=shape), 1.2).astype(np.float32) * 100
    zz, yy, xx = np.indices(shape, dtype=np.float32)
    moving = sndi.map_coordinates(fixed, [zz - 0.5, yy + 0.5, xx - 1.5], order=1, mode="nearest").astype(np.float32)
    grid = tuple((n - p) // s + 1 for n, p, s in zip(shape, patch, step))
    flow = np.zeros((3,) + grid, dtype=np.float32)
    flow[0], flow[1], flow[2] = 1.0, -1.0, 1.0
    flow[:, 0, 1, 2] = np.nan
    flow[:, 2, 3, 4] = np.nan
    refined, count = sf._refine_flow_vectors_subpixel(flow, moving, fixed, patch_size_zyx=patch, step_zyx=step, config=cfg)
    np.savez_compressed(
        HERE / "sofima_glue_jax.npz",
        base=base, residual=res, stride=np.asarray(stride), box_start=np.asarray(box), composed=composed,
        fixed=fixed, moving=moving, patch=np.asarray(patch), step=np.asarray(step), flow=flow,
        refined=refined, refined_count=np.int64(count),
    )


def synthetic_barcode_table(rng, n=4000, n_tiles=3):
    """Decoded-transcript table with the columns the post-decode stage reads; blanks are dimmer,
    smaller and further from their codeword on average, with overlap so the filter has work."""
    genes = np.array([f"gene{i:03d}" for i in range(130)] + [f"Blank-{i:02d}" for i in range(10)])
    blank = rng.uniform(size=n) < 0.12
    gid = np.where(blank, rng.integers(130, 140, size=n), rng.integers(0, 130, size=n))
    mag = np.where(blank, rng.gamma(3.0, 0.25, size=n), rng.gamma(6.0, 0.3, size=n)).astype(np.float32)
    area = np.where(blank, rng.integers(16, 30, size=n), rng.integers(16, 90, size=n)).astype(np.float64)
    dist = np.where(blank, rng.uniform(0.25, 0.5176, size=n), rng.uniform(0.02, 0.5176, size=n) ** 1.3).astype(np.float32)
    tile = rng.integers(0, n_tiles, size=n)
    # tiles overlap in y; some transcripts are duplicated across tiles and across planes
    zyx = np.stack([np.round(rng.uniform(0, 12, size=n)) * 0.31, rng.uniform(0, 60, size=n) + tile * 45.0,
                    rng.uniform(0, 60, size=n)], axis=1)
    dup = rng.choice(n, size=n // 10, replace=False)
    src = rng.choice(np.setdiff1d(np.arange(n), dup), size=dup.size, replace=False)
    zyx[dup] = zyx[src] + rng.normal(0, 0.03, size=(dup.size, 3)) + np.array([0.31, 0, 0]) * rng.integers(0, 2, size=(dup.size, 1))
    gid[dup] = gid[src]
    same_tile = rng.uniform(size=dup.size) < 0.5
    tile[dup] = np.where(same_tile, tile[src], (tile[src] + 1) % n_tiles)
    mag[7] = np.nan
    return pd.DataFrame({"gene_id": genes[gid], "magnitude_mean": mag, "area": area, "distance_min": dist,
                         "tile_idx": tile, "global_z": zyx[:, 0], "global_y": zyx[:, 1], "global_x": zyx[:, 2]}), genes


def make_postdecode(pdm):
    """_filter_all_barcodes_blank_fraction, _remove_duplicates_* (PixelDecoder.py:3386-3846, 4137-4363)
    and calculate_F1_with_radius (cli/statphysbio_simulation/calculate_F1.py:19-200) on a synthetic table."""
    import importlib

    f1 = importlib.import_module("merfish3danalysis.cli.statphysbio_simulation.calculate_F1")
    rng = np.random.default_rng(2024)
    table, genes = synthetic_barcode_table(rng)
    out = {"gene_names": genes, "gene_code": np.searchsorted(np.sort(genes), table["gene_id"].to_numpy()),
           "gene_sorted": np.sort(genes)}
    for c in table.columns[1:]:
        out[c] = table[c].to_numpy()

    def run_filter(tag, df, **kw):
        dec = pdm.PixelDecoder.__new__(pdm.PixelDecoder)
        dec._df_barcodes_loaded = df.copy()
        dec._blank_count, dec._barcode_count, dec._verbose = 10, 140, 0
        dec._filter_all_barcodes_blank_fraction(**kw)
        d = dec._blank_fraction_filter_results
        kept = dec._df_filtered_barcodes
        out[f"{tag}_kept_rows"] = kept.index.to_numpy()
        out[f"{tag}_kept_fraction"] = kept["blank_fraction"].to_numpy()
        out[f"{tag}_kept_bin"] = kept["blank_fraction_bin"].to_numpy()
        for k in ("chosen_threshold", "achieved_gross_misid_rate", "target_reached", "all_histogram", "blank_histogram",
                  "blank_fraction_histogram", "intensity_bins", "voxel_number_bins", "vector_distance_bins"):
            out[f"{tag}_{k}"] = np.asarray(d[k])
        out[f"{tag}_sweep"] = d["threshold_sweep"].to_numpy(dtype=float)
        out[f"{tag}_reason"] = np.asarray(d.get("reason", ""))
        return kept

    kept = run_filter("auto", table, target_gross_misid_rate=0.05)
    run_filter("strict", table, target_gross_misid_rate=1e-4)
    run_filter("explicit", table, target_gross_misid_rate=0.1, intensity_bins=[0.0, 0.8, 1.5, 2.5, 10.0],
               voxel_number_bins=[15.5, 20.5, 40.5, 100.5], vector_distance_bins=np.linspace(0.0, 0.52, 6))
    few = table.iloc[:40].copy()
    few["area"] = 16.0 + (np.arange(40) % 4)
    out["few_area"] = few["area"].to_numpy()
    run_filter("few", few, target_gross_misid_rate=0.05)
    noblank = table[~table["gene_id"].str.startswith("Blank")].iloc[:50]
    out["noblank_rows"] = noblank.index.to_numpy()
    run_filter("noblank", noblank, target_gross_misid_rate=0.05)

    # duplicate removal on the auto-filtered table: 2-D style within-tile collapse, then tile overlap
    dec = pdm.PixelDecoder.__new__(pdm.PixelDecoder)
    dec._verbose = 0
    dec._df_filtered_barcodes = kept.copy()
    dec._df_filtered_barcodes["row_id"] = kept.index.to_numpy()
    dec._remove_duplicates_within_tile(radius_xy=0.098, radius_z=0.31)
    out["within_rows"] = dec._df_filtered_barcodes["row_id"].to_numpy()
    dec._remove_duplicates_in_tile_overlap(radius=0.75)
    out["overlap_rows"] = dec._df_filtered_barcodes["row_id"].to_numpy()

    # F1 scorer: ground truth = a jittered subset of the kept transcripts + unmatched extras
    final = dec._df_filtered_barcodes
    q = final[["global_z", "global_y", "global_x"]].to_numpy()
    qg = final["gene_id"].to_numpy()
    pick = rng.uniform(size=len(final)) < 0.8
    gt = q[pick] + rng.normal(0, 0.25, size=(int(pick.sum()), 3))
    gtg = qg[pick].copy()
    swap = rng.uniform(size=gtg.size) < 0.05
    gtg[swap] = genes[rng.integers(0, 130, size=int(swap.sum()))]
    extra = rng.uniform((0, 0, 0), (4, 150, 60), size=(200, 3))
    gt = np.concatenate([gt, extra])
    gtg = np.concatenate([gtg, genes[rng.integers(0, 130, size=200)]])
    out["f1_gt"], out["f1_gt_gene"] = gt, gtg.astype(str)
    for tag, radius in (("r1", 1.0), ("r03", 0.3)):
        res, tp, fp, fn = f1.calculate_F1_with_radius(q, qg, gt, gtg, radius)
        out[f"f1_{tag}_metrics"] = np.asarray([res["F1 Score"], res["Precision"], res["Recall"], res["True Positives"],
                                               res["False Positives"], res["False Negatives"]], dtype=float)
        out[f"f1_{tag}_tp"], out[f"f1_{tag}_fp"], out[f"f1_{tag}_fn"] = tp, fp, fn
    np.savez_compressed(HERE / "postdecode.npz", **out)


def make_postdecode_lr(pdm):
    """_filter_all_barcodes_LR (PixelDecoder.py:3907-4057) on the synthetic table + shape features."""
    rng = np.random.default_rng(2024)
    table, _genes = synthetic_barcode_table(rng)
    rng2 = np.random.default_rng(99)
    n = len(table)
    blank = table["gene_id"].str.startswith("Blank").to_numpy()
    table["magnitude_mean"] = table["magnitude_mean"].fillna(1.0)
    table["signal_mean"] = (table["magnitude_mean"] * rng2.uniform(0.8, 1.2, n)).astype(np.float32)
    table["s-b_mean"] = (table["signal_mean"] - np.where(blank, 0.5, 0.2) * rng2.uniform(0.5, 1.5, n)).astype(np.float32)
    for k, scale in enumerate((4.0, 2.0, 0.8)):
        table[f"inertia_tensor_eigvals-{k}"] = rng2.gamma(2.0, scale, n) * np.where(blank, 0.7, 1.0)
    out = {c: table[c].to_numpy() for c in table.columns if c != "gene_id"}
    out["is_blank"] = blank
    for tag, is3d, target in (("3d", True, 0.05), ("2d", False, 0.2)):
        dec = pdm.PixelDecoder.__new__(pdm.PixelDecoder)
        dec._df_barcodes_loaded = table.copy()
        dec._blank_count, dec._barcode_count, dec._verbose, dec._is_3D = 10, 140, 0, is3d
        dec._filter_all_barcodes_LR(lr_fdr_target=target)
        out[f"{tag}_kept_rows"] = dec._df_filtered_barcodes.index.to_numpy()
        out[f"{tag}_probability"] = dec._df_filtered_barcodes["predicted_probability"].to_numpy()
    few = table[~blank].iloc[:30]
    dec = pdm.PixelDecoder.__new__(pdm.PixelDecoder)
    dec._df_barcodes_loaded = few.copy()
    dec._blank_count, dec._barcode_count, dec._verbose, dec._is_3D = 10, 140, 0, True
    dec._filter_all_barcodes_LR(lr_fdr_target=0.05)
    out["noblank_kept_rows"] = dec._df_filtered_barcodes.index.to_numpy()
    np.savez_compressed(HERE / "postdecode_lr.npz", **out)


def main():
    rl, mv, dw, sf, pdm = import_reference()
    if "--postdecode-lr-only" in sys.argv:
        make_postdecode_lr(pdm)
        return
    if "--sofima-jax-only" in sys.argv:
        make_sofima_jax_glue(sf)
        return
    if "--postdecode-only" in sys.argv:
        make_postdecode(pdm)
        return
    if "--extract-only" in sys.argv:
        make_extract(pdm)
        return
    if "--optimize-only" in sys.argv:
        make_optimize(pdm)
        return
    if "--decode-only" in sys.argv:
        make_decode(pdm)
        make_extract(pdm)
        return
    make_rlgc(rl)
    make_decode(pdm)
    make_extract(pdm)
    make_optimize(pdm)
    make_normalization(pdm)
    make_registration(mv, dw, sf)
    make_sofima_jax_glue(sf)
    make_postdecode(pdm)
    make_postdecode_lr(pdm)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
