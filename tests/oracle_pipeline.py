"""Test infrastructure: the simulation pipeline (preprocess -> decode -> F1) run through the ORACLE on the CPU, to stand
next to the HIP pipeline in the F1 matrix (tests/golden/make_golden_f1_matrix.py, tests/test_gpu_f1_matrix.py).

Every device stage is the oracle's NumPy / SciPy restatement:
  preprocess   RLGC of fiducial and readout images (``oracle.rlgc_ref``, the Philox split stream of the build),
               ``register_pair_to_fixed`` (``oracle.registration_ref``), the SOFIMA residual flow
               (``oracle.sofima_ref``)                                   - DataRegistration.py:390-693, 906-1121;
  decode       readout x predictor, affine / affine + flow warp, Gaussian low-pass, global normalisation vectors, pixel
               decode, connected components + transcript table (``oracle.decode_ref``, ``oracle.ccl_ref``)
                                                                          - PixelDecoder.py:981-1199, 1828-2041, 2354-3201.
What is NOT restated a second time is host-side table logic, which has no device arithmetic in it and is pinned by
reference-run fixtures of its own: the optimise-normalisation loop and iterative vectors (a22,
optimize_normalization.npz), the tile drivers (a23), blank-fraction filtering and de-duplication (f-1, postdecode.npz),
the F1 scorer (postdecode.npz).  ``OracleDecoder`` inherits those from the package's ``PixelDecoder`` and overrides
every method that touches the GPU.
"""

from __future__ import annotations

import gc
import warnings

import numpy as np
import pandas as pd
import scipy.ndimage as sndi
from scipy.ndimage import gaussian_filter

from conftest import pkg
from oracle import ccl_ref, decode_ref, registration_ref, rlgc_ref, sofima_ref

# the F1 matrix of the reference's acceptance test (tests/test_simulation_example_pipeline.py:48-52, 1244-1318): axial
# spacing -> F1 search radius; each cell is run affine-only and with the SOFIMA residual flow
# decode settings per cell as the reference's matrix chooses them (tests/test_simulation_example_pipeline.py:39-47):
# magnitude threshold (0.9, 10) for Nyquist-sampled 3-D data, (0.7, 10) / (0.2, 10) for per-plane decoding at 3 x / 5 x
# the Nyquist step, minimum pixels 7 in 2-D.  In 3-D the reference's 28 voxels is sized for its real PSF; the
# synthetic spots here (sigma 1.2 x 1.5 x 1.5 px) cover fewer: 9.
CELLS = {
    "dz0.315": dict(dz=0.315, radius=1.0, seed=21, decode=dict(minimum_pixels=9, magnitude_threshold=(0.9, 10.0)),
                    # lateral shifts that round to 1..4 px: the range bench config 3 draws and the one in which
                    # phase_cross_correlation(disambiguate=True) meets its three-sample corner tiles
                    shifts=[(0, 0, 0), (0.4, 2.6, -3.2), (-0.3, -1.4, 3.7), (0.2, 3.3, 1.8), (0.0, -2.9, -3.6),
                            (0.5, 1.2, 3.9), (-0.4, -3.8, -1.6), (0.1, 2.9, -2.2)]),
    "dz1.0": dict(dz=1.0, radius=1.0, seed=22, decode=dict(minimum_pixels=7, magnitude_threshold=(0.7, 10.0)),
                  shifts=[(0, 0, 0), (0.3, 6.4, -7.1), (-0.2, -7.7, 6.2), (0.2, 8.6, 5.8), (0.0, -6.9, -8.4),
                          (0.3, 5.2, 9.3), (-0.3, -8.8, -5.6), (0.1, 6.9, -7.7)]),
    "dz1.5": dict(dz=1.5, radius=1.5, seed=23, decode=dict(minimum_pixels=7, magnitude_threshold=(0.2, 10.0)),
                  shifts=[(0, 0, 0), (0.2, 5.4, -6.1), (-0.2, -6.7, 5.2), (0.1, 7.6, 4.8), (0.0, -5.9, -7.4),
                          (0.2, 4.2, 8.3), (-0.2, -7.8, -4.6), (0.1, 5.9, -6.7)]),
}
SHAPE = (16, 256, 256)


def make_cell(root, name):
    """The cell's dataset in the statphysbio on-disk layout -> (in-memory datastore, ground-truth table)."""
    sim = pkg("simulation")
    c = CELLS[name]
    path = sim.make_simulation(root / name, shape_zyx=SHAPE, n_rna=3000, n_beads=700, round_shifts_px=c["shifts"],
                               deformation_px=1.2, seed=c["seed"], voxel_zyx_um=(c["dz"], 0.098, 0.098))
    store = sim.convert_data(sim.convert_simulation(path))
    t = store.tile_ids[0]
    store.set_global_coord_xforms_um(t, np.eye(4, dtype=np.float32), np.zeros(3, dtype=np.float32),
                                     np.asarray(store.voxel_size_zyx_um, dtype=np.float32))
    return store, pd.read_csv(path / "GT_spots.csv")


def predictor(image):
    """Stand-in for U-FISH (third party, out of scope), shared by both pipelines: difference-of-Gaussians spot likelihood
    in [0, 1] and its strongest voxels as the localisation table."""
    img = np.asarray(image, dtype=np.float32)
    dog = gaussian_filter(img, (1.0, 1.2, 1.2)) - gaussian_filter(img, (2.5, 3.5, 3.5))
    dog = np.clip(dog, 0, None)
    prob = np.clip(dog / max(float(np.percentile(dog, 99.9)), 1e-6), 0.0, 1.0).astype(np.float32)
    flat = np.argsort(prob.ravel())[-20:]
    z, y, x = np.unravel_index(flat, prob.shape)
    return pd.DataFrame({"axis-0": z, "axis-1": y, "axis-2": x}), prob


# ---------------------------------------------------------------------------------------------------- preprocess
def _decon_u16(image, psf):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = rlgc_ref.rlgc(np.asarray(image), np.asarray(psf, dtype=np.float32), rng="philox", rng_seed=42)
    return np.clip(out, 0, 65535).astype(np.uint16)      # DataRegistration.py:435, 993-995


def oracle_preprocess(store, deformable: bool, fft_workers: int = 8, progress=None) -> None:
    """DataRegistration.register_all_tiles on the CPU: deconvolved fiducials, round transforms, (SOFIMA flow fields,)
    deconvolved readouts and predictor images written into ``store`` (one tile)."""
    t = store.tile_ids[0]
    spacing = tuple(float(v) for v in store.voxel_size_zyx_um)
    psfs = store.channel_psfs
    rlgc_ref.FFT_WORKERS = fft_workers
    try:
        fid = []
        for r, rid in enumerate(store.round_ids):          # :390-453
            raw = store.load_local_corrected_image(tile=t, round=rid, return_future=False)
            fid.append(_decon_u16(raw, psfs[0]))
            store.save_local_deconvolved_fiducial_image(fid[-1], tile=t, round=rid)
            if progress:
                progress(f"fiducial {rid}")
        store.save_local_round_transform_zyx_um(np.eye(4, dtype=np.float32), tile=t, round=store.round_ids[0])
        for r, rid in enumerate(store.round_ids[1:], start=1):   # :456-558
            T = registration_ref.register_pair_to_fixed(fid[0], fid[r], spacing)
            store.save_local_round_transform_zyx_um(np.asarray(T, dtype=np.float32), tile=t, round=rid)
        if deformable:                                       # :561-693
            fixed = fid[0].astype(np.float32)
            for r, rid in enumerate(store.round_ids[1:], start=1):
                T = store.load_local_round_transform_zyx_um(tile=t, round=rid)
                warped = registration_ref.warp_array_to_reference(fid[r], T, spacing, fixed.shape)
                flow, md = sofima_ref.estimate_sofima_flow_field_xyz_px(fixed, warped.astype(np.float32))
                store.save_local_sofima_flow_field(
                    flow, tile=t, round=rid, reference_round=store.round_ids[0],
                    map_stride_zyx_px=md["map_stride_zyx_px"], map_box_start_xyz_px=md["map_box_start_xyz_px"],
                    map_box_size_xyz_px=md["map_box_size_xyz_px"], reference_shape_zyx_px=tuple(fixed.shape),
                    moving_shape_zyx_px=tuple(warped.shape), sofima_status=md.get("status", "ok"),
                    valid_flow_vectors=md.get("valid_flow_vectors"), return_future=False)
                if progress:
                    progress(f"sofima {rid}: {md.get('status')}, {md.get('valid_flow_vectors')} vectors")
        for bit_id in store.bit_ids:                         # :906-1121
            ex, _em = store.load_local_wavelengths_um(tile=t, bit=bit_id)
            raw = store.load_local_corrected_image(tile=t, bit=bit_id, return_future=False)
            decon = _decon_u16(raw, psfs[1 if ex < 0.600 else 2])
            store.save_local_deconvolved_readout_image(decon, tile=t, bit=bit_id)
            _spots, prob = predictor(decon)
            store.save_local_feature_predictor_image(prob, tile=t, bit=bit_id)
            if progress:
                progress(f"readout {bit_id}")
    finally:
        rlgc_ref.FFT_WORKERS = 0


# ---------------------------------------------------------------------------------------------------- decode
def warp_with_affine_and_flow(image, transform_zyx_um, spacing, reference_shape, flow_xyz, stride_zyx, box_start_xyz,
                              origin=(0.0, 0.0, 0.0), mode="constant", cval=0.0):
    """multiview_registration.py:905-1131 (order 1): the flow is sampled at (index - box start) / stride with the
    image's boundary mode, the displaced reference coordinate goes to microns, through the affine and back to pixels,
    and the image is sampled once.  Pinned by tests/golden/registration.npz (tests/test_oracle_golden.py)."""
    image = np.asarray(image, dtype=np.float32)
    flow = np.asarray(flow_xyz, dtype=np.float32)
    sp = np.asarray(spacing, dtype=np.float64)
    org = np.asarray(origin, dtype=np.float64)
    T = np.asarray(transform_zyx_um, dtype=np.float64)
    idx = np.indices(tuple(int(v) for v in reference_shape), dtype=np.float64)
    start_zyx = np.asarray(box_start_xyz, dtype=np.float64)[::-1]
    fc = np.stack([(idx[a] - start_zyx[a]) / float(stride_zyx[a]) for a in range(3)])
    d_zyx = [sndi.map_coordinates(flow[c], fc, order=1, mode=mode, cval=cval) for c in (2, 1, 0)]
    phys = [(idx[a] + d_zyx[a]) * sp[a] + org[a] for a in range(3)]
    src = [(T[a, 0] * phys[0] + T[a, 1] * phys[1] + T[a, 2] * phys[2] + T[a, 3] - org[a]) / sp[a] for a in range(3)]
    return sndi.map_coordinates(image, np.stack(src), order=1, mode=mode, cval=cval).astype(np.float32)


def _oracle_decoder_class():
    PD = pkg("PixelDecoder")
    dw = pkg("utils.decode_warping")

    class OracleDecoder(PD.PixelDecoder):
        """Host control flow of the package, device stages from the oracle (see the module docstring)."""

        def _weighted_warped_bit(self, tile_id, bit_id, gpu_id: int = 0):      # :1047-1070, decode_warping.py:184-245
            ds = self._datastore
            readout = np.asarray(ds.load_local_readout_image(tile=tile_id, bit=bit_id, return_future=False))
            pred = np.asarray(ds.load_local_feature_predictor_image(tile=tile_id, bit=bit_id, return_future=False))
            _ex, em = ds.load_local_wavelengths_um(tile=tile_id, bit=bit_id)
            weighted = readout.astype(np.float32) * pred.astype(np.float32)
            round_id, rt = dw.load_bit_round_transform_zyx_um(ds, tile=tile_id, bit_id=bit_id)
            T = dw.compose_decode_warp_transform_zyx_um(
                round_transform_zyx_um=rt, chromatic_transform_zyx_um=ds.load_chromatic_affine_transform_zyx_um(
                    wavelength_um=em))
            loaded = ds.load_local_sofima_flow_field(tile=tile_id, round=round_id, return_future=False) \
                if round_id is not None else None
            spacing = ds.voxel_size_zyx_um
            if loaded is None and np.allclose(T, np.eye(4, dtype=np.float32)):
                return weighted, em
            if loaded is not None and not str(loaded[1].get("sofima_status", "")).startswith("identity_fallback"):
                flow, attrs = loaded
                return warp_with_affine_and_flow(weighted, T, spacing, attrs["reference_shape_zyx_px"], flow,
                                                 attrs["map_stride_zyx_px"], attrs["map_box_start_xyz_px"]), em
            return registration_ref.warp_array_to_reference(weighted, T, spacing, weighted.shape).astype(np.float32), em

        def _load_bit_data(self, feature_predictor_threshold=0.1, gpu_id: int = 0) -> None:   # :1828-1946
            bit_ids = self._datastore.bit_ids[0: self._n_merfish_bits]
            images, self._em_wvl = [], []
            for bit_id in bit_ids:
                warped, em = self._weighted_warped_bit(self._tile_idx, bit_id, gpu_id)
                images.append(warped[self._z_slice])
                self._em_wvl.append(em)
            self._image_data = np.stack(images, axis=0)
            voxel = self._datastore.voxel_size_zyx_um
            self._pixel_size, self._axial_step = voxel[1], voxel[0]
            stage = self._datastore.load_local_stage_position_zyx_um(tile=self._tile_idx, round=0)
            stage_origin, c2s = (None, np.eye(4, dtype=np.float32)) if stage is None else stage
            affine, origin, spacing = self._datastore.load_global_coord_xforms_um(tile=self._tile_idx)
            if affine is None or origin is None or spacing is None:
                affine = np.eye(4)
                origin = np.zeros(3, dtype=np.float32) if stage_origin is None else stage_origin
                spacing = voxel
            self._affine = np.asarray(affine, dtype=np.float32)
            self._origin = np.asarray(origin, dtype=np.float32)
            self._spacing = np.asarray(spacing, dtype=np.float32)
            self._camera_to_stage_affine = np.asarray(c2s, dtype=np.float32)

        def _lowpass_image(self, image, sigma):
            if sigma is None or np.any(np.asarray(sigma, dtype=float) == 0):
                return image
            return decode_ref.lowpass(np.asarray(image)[None], sigma, self._is_3D)[0]

        def _lp_filter(self, gpu_id: int = 0, sigma=PD.DEFAULT_DECODE_LOWPASS_SIGMA) -> None:   # :1982-2024
            self._image_data_lp = decode_ref.lowpass(self._image_data, sigma, self._is_3D)
            self._filter_type = "lp"
            del self._image_data

        def _global_normalization_vectors(self, low_percentile_cut=10.0, high_percentile_cut=90.0,
                                          hot_pixel_threshold=50000, gpu_id=0, tile_indices=None,
                                          lowpass_sigma=PD.DEFAULT_DECODE_LOWPASS_SIGMA) -> None:   # :981-1199
            sigma = self._effective_lowpass_sigma(lowpass_sigma)
            ds = self._datastore
            tiles = [ds.tile_ids[i] for i in tile_indices] if tile_indices is not None else list(ds.tile_ids)[:5]
            n_all = len(ds.bit_ids)
            norm, bkd = np.ones(n_all, dtype=np.float32), np.zeros(n_all, dtype=np.float32)
            for b, bit_id in enumerate(ds.bit_ids):
                images = []
                for tile_id in tiles:
                    img, _em = self._weighted_warped_bit(tile_id, bit_id, gpu_id)
                    img = np.array(img, dtype=np.float32, copy=True)
                    img[img > hot_pixel_threshold] = np.median(img[img.shape[0] // 2])
                    images.append(self._lowpass_image(img[self._z_slice], sigma))
                low = [im[im < np.percentile(im, low_percentile_cut)] for im in images]
                low = np.concatenate([v.ravel() for v in low]) if low else np.zeros(0, np.float32)
                bkd[b] = np.median(low) if low.size else 0
                high = []
                for im in images:
                    v = np.maximum(im - bkd[b], np.float32(0))
                    high.append(v[v > np.percentile(v, high_percentile_cut)].ravel())
                high = np.concatenate(high) if high else np.zeros(0, np.float32)
                norm[b] = np.median(high) if high.size else 1
            ds.save_decode_normalization_vectors(self._decode_run_key, "global", norm, bkd,
                                                 decode_mode=self._effective_decode_mode)
            self._global_background_vector, self._global_normalization_vector = bkd, norm
            self._global_normalization_loaded = True

        def _decode_pixels(self, magnitude_threshold=(1.1, 2.0), gpu_id: int = 0) -> None:    # :2523-2643
            stack = self._image_data_lp if self._filter_type == "lp" else self._image_data
            bkd = nrm = None
            if self._iterative_normalization_loaded:
                bkd, nrm = self._iterative_background_vector, self._iterative_normalization_vector
            elif self._global_normalization_loaded:
                bkd, nrm = self._global_background_vector, self._global_normalization_vector
            dec, mag, dist, scaled = decode_ref.decode_pixels(
                stack, np.asarray(self._decoding_matrix), self._pixel_assignment_threshold, tuple(magnitude_threshold),
                background=None if bkd is None else np.asarray(bkd, dtype=np.float32)[: stack.shape[0]],
                normalization=None if nrm is None else np.asarray(nrm, dtype=np.float32)[: stack.shape[0]],
                excluded_indices=tuple(self._excluded_codeword_indices), arith="device")
            self._results = {"decoded": dec, "magnitude": mag, "distance": dist, "scaled": scaled}

        def _extract_barcodes(self, minimum_pixels: int = 3, maximum_pixels: int = 500, gpu_id: int = 0) -> None:
            intensity = (self._image_data_lp if self._filter_type == "lp" else self._image_data) \
                if self._optimize_normalization_weights else self._results["scaled"]       # :2935-2941
            self._df_barcodes = ccl_ref.extract_barcodes(
                self._results["decoded"], intensity, self._results["magnitude"], self._results["distance"],
                self._codebook_matrix, self._gene_ids, self._transcript_distance_threshold, is_3d=self._is_3D,
                minimum_pixels=int(minimum_pixels), maximum_pixels=int(maximum_pixels), tile_idx=self._tile_idx,
                spacing=self._spacing, origin=self._origin, affine=self._affine,
                camera_to_stage=self._camera_to_stage_affine)

        def _cleanup(self) -> None:
            for name in ("_image_data_lp", "_image_data"):
                if hasattr(self, name):
                    delattr(self, name)
            self._results = {}
            self._df_barcodes = pd.DataFrame()
            gc.collect()

        def decode_one_tile(self, tile_idx=0, gpu_id=0, display_results=False, return_results=False,
                            lowpass_sigma=PD.DEFAULT_DECODE_LOWPASS_SIGMA, magnitude_threshold=None,
                            minimum_pixels=None, use_normalization=True, normalization_method=None,
                            feature_predictor_threshold=0.1):                               # :4471-4579
            if magnitude_threshold is None:
                magnitude_threshold = PD.DEFAULT_DECODE_MAGNITUDE_THRESHOLD
            if minimum_pixels is None:
                minimum_pixels = self._default_minimum_pixels()
            self._prepare_normalization_state(normalization_method=normalization_method,
                                              use_normalization=use_normalization, gpu_id=gpu_id,
                                              lowpass_sigma=lowpass_sigma)
            self._tile_idx = tile_idx
            self._load_bit_data(feature_predictor_threshold=feature_predictor_threshold, gpu_id=gpu_id)
            self._filter_type = "raw"
            sigma = self._effective_lowpass_sigma(lowpass_sigma)
            if sigma is not None and not np.any(np.asarray(sigma, dtype=float) == 0):
                self._lp_filter(sigma=sigma, gpu_id=gpu_id)
            self._decode_pixels(magnitude_threshold=magnitude_threshold, gpu_id=gpu_id)
            self._extract_barcodes(minimum_pixels=minimum_pixels, gpu_id=gpu_id)
            return None

    return OracleDecoder


def oracle_decode_and_score(store, gt, radius: float, decode_kw: dict) -> dict:
    """optimize_normalization_by_decoding (one tile, three iterations: the simulation CLI's setting,
    cli/statphysbio_simulation/pixeldecode.py:273-281) -> decode_all_tiles -> calculate_F1, all on the CPU."""
    sim = pkg("simulation")
    dec = _oracle_decoder_class()(store, merfish_bits=16, num_gpus=1, verbose=0)
    dec.optimize_normalization_by_decoding(n_random_tiles=1, n_iterations=3, **decode_kw)
    dec.decode_all_tiles(assign_to_cells=False, normalization_method="iterative", **decode_kw)
    return {**sim.calculate_F1(store, gt, search_radius=radius), **_details(store, dec)}


def _details(store, dec) -> dict:
    """What the F1 rests on, for diagnosis when two pipelines drift apart: transcripts before and after the
    blank-fraction filter, the filter's threshold, the iterative normalisation vectors."""
    norm, bkd = store.load_decode_normalization_vectors(None, "iterative")
    diag = getattr(dec, "_blank_fraction_filter_results", {}) or {}
    scalars = {k: (float(v) if isinstance(v, (float, np.floating)) else int(v)) for k, v in diag.items()
               if isinstance(v, (float, int, np.floating, np.integer)) and not isinstance(v, bool)}
    return {"decoded_before_filter": int(len(dec._df_barcodes_loaded)),
            "decoded_after_filter": int(len(dec._df_filtered_barcodes)),
            "filter": scalars, "normalization": np.round(np.asarray(norm, dtype=float), 1).tolist(),
            "background": np.round(np.asarray(bkd, dtype=float), 1).tolist()}


def hip_pipeline_score(store, gt, radius: float, deformable: bool, decode_kw: dict) -> dict:
    """The same pipeline through the package's operators on the GPU."""
    sim, DR, PD = pkg("simulation"), pkg("DataRegistration"), pkg("PixelDecoder")
    reg = DR.DataRegistration(store, decon_fiducial=True, decon_readout=True,
                              perform_deformable_registration=deformable, num_gpus=1, verbose=0,
                              feature_predictor=predictor)
    reg.register_all_tiles()
    dec = PD.PixelDecoder(store, merfish_bits=16, num_gpus=1, verbose=0)
    dec.optimize_normalization_by_decoding(n_random_tiles=1, n_iterations=3, **decode_kw)
    dec.decode_all_tiles(assign_to_cells=False, normalization_method="iterative", **decode_kw)
    return {**sim.calculate_F1(store, gt, search_radius=radius), **_details(store, dec)}


def hip_readouts_then_score(store, gt, transforms_zyx_um, radius: float, decode_kw: dict,
                            reference_stream: bool = False) -> dict:
    """The HIP pipeline downstream of GIVEN round transforms (tests/test_gpu_f1_reference_run.py): readout
    deconvolution (``utils.rlgc`` in the call form of DataRegistration.py:379-387, clip -> uint16), the shared predictor
    stand-in, then ``PixelDecoder`` (optimise normalisation, decode all tiles, blank-fraction filter) and the scorer.

    ``reference_stream``: the photon splits of every iteration are the ones the reference run drew (NumPy's
    ``default_rng(42).binomial``, replayed by the oracle in spawned worker processes and handed to the HIP solver through
    its injection hook) instead of the build's Philox stream - the two pipelines then differ by float rounding only.
    Returns the scorer's dictionary plus ``uint16_flips`` (fraction of deconvolved voxels that differ from the replay's,
    reference_stream only)."""
    import torch

    sim, PD, rl = pkg("simulation"), pkg("PixelDecoder"), pkg("utils.rlgc")
    t = store.tile_ids[0]
    psfs = store.channel_psfs
    for rid, T in zip(store.round_ids, transforms_zyx_um):
        store.save_local_round_transform_zyx_um(np.asarray(T, dtype=np.float32), tile=t, round=rid)
    raws, bit_psfs = [], []
    for bit_id in store.bit_ids:
        ex, _em = store.load_local_wavelengths_um(tile=t, bit=bit_id)
        raws.append(np.asarray(store.load_local_corrected_image(tile=t, bit=bit_id, return_future=False)))
        bit_psfs.append(np.asarray(psfs[1 if ex < 0.600 else 2], dtype=np.float32))
    replays = None
    if reference_stream:
        import multiprocessing as mp
        import os
        from concurrent.futures import ProcessPoolExecutor

        from oracle_workers import rlgc_reference_stream

        n_proc = max(1, min(8, (os.cpu_count() or 2) // 2))
        with ProcessPoolExecutor(n_proc, mp_context=mp.get_context("spawn")) as pool:   # children never touch the GPU
            replays = list(pool.map(rlgc_reference_stream, [(raws[b], bit_psfs[b], 42, 2) for b in range(len(raws))]))
    flips = total = 0
    for b, bit_id in enumerate(store.bit_ids):
        if replays is None:
            out = rl.chunked_rlgc(image=raws[b], psf=bit_psfs[b], gpu_id=0, crop_yx=2048, crop_z=None,
                                  release_memory=False)
        else:
            ref, r_iters, r_stop, splits = replays[b]
            o, info = rl.rlgc_device(torch.from_numpy(raws[b]).to("cuda:0"), torch.from_numpy(bit_psfs[b]).to("cuda:0"),
                                     injected_splits=torch.from_numpy(splits.astype(np.float32)).to("cuda:0"))
            assert (info.iterations, info.stop_reason) == (r_iters, r_stop), (bit_id, info.iterations, info.stop_reason,
                                                                               r_iters, r_stop)
            out = o.cpu().numpy()
            d = np.abs(np.clip(out, 0, 65535).astype(np.uint16).astype(np.int64)
                       - np.clip(ref, 0, 65535).astype(np.uint16).astype(np.int64))
            assert d.max() <= 1, (bit_id, int(d.max()))
            flips += int((d > 0).sum())
            total += d.size
        decon = np.clip(out, 0, 65535).astype(np.uint16)
        store.save_local_deconvolved_readout_image(decon, tile=t, bit=bit_id)
        _spots, prob = predictor(decon)
        store.save_local_feature_predictor_image(prob, tile=t, bit=bit_id)
    dec = PD.PixelDecoder(store, merfish_bits=16, num_gpus=1, verbose=0)
    dec.optimize_normalization_by_decoding(n_random_tiles=1, n_iterations=3, **decode_kw)
    dec.decode_all_tiles(assign_to_cells=False, normalization_method="iterative", **decode_kw)
    res = {**sim.calculate_F1(store, gt, search_radius=radius), **_details(store, dec)}
    if total:
        res["uint16_flips"] = flips / total
    return res


def oracle_readouts_then_score(store, gt, transforms_zyx_um, radius: float, decode_kw: dict, fft_workers: int = 8) -> dict:
    """The ORACLE pipeline downstream of given round transforms (the twin of ``hip_readouts_then_score``)."""
    t = store.tile_ids[0]
    psfs = store.channel_psfs
    for rid, T in zip(store.round_ids, transforms_zyx_um):
        store.save_local_round_transform_zyx_um(np.asarray(T, dtype=np.float32), tile=t, round=rid)
    rlgc_ref.FFT_WORKERS = fft_workers
    try:
        for bit_id in store.bit_ids:
            ex, _em = store.load_local_wavelengths_um(tile=t, bit=bit_id)
            raw = store.load_local_corrected_image(tile=t, bit=bit_id, return_future=False)
            decon = _decon_u16(raw, psfs[1 if ex < 0.600 else 2])
            store.save_local_deconvolved_readout_image(decon, tile=t, bit=bit_id)
            _spots, prob = predictor(decon)
            store.save_local_feature_predictor_image(prob, tile=t, bit=bit_id)
    finally:
        rlgc_ref.FFT_WORKERS = 0
    return oracle_decode_and_score(store, gt, radius, decode_kw)
