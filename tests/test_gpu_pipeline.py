"""GPU: the two operators chained through the datastore duck type on a small synthetic experiment -
DataRegistration (fiducial deconvolution + round registration + readout deconvolution + predictor) then
PixelDecoder (global normalisation, decode_all_tiles with registration applied at decode time, filter,
de-duplication) - scored against the planted emitters with the simulation scorer.  Catches interface
mismatches that per-stage parity tests cannot (DataRegistration.py:1522-1591 -> PixelDecoder.py:4759-4870)."""

import numpy as np
import pandas as pd
import pytest
from scipy.ndimage import gaussian_filter, shift as ndi_shift

from conftest import pkg
from helpers import codebook_frame

pytestmark = pytest.mark.gpu
SHAPE = (16, 128, 128)
VOXEL = (0.31, 0.1, 0.1)
N_BITS, N_ROUNDS = 16, 8


def _codebook(seed=0):
    import itertools

    rng = np.random.default_rng(seed)
    combos = list(itertools.combinations(range(N_BITS), 4))
    rng.shuffle(combos)
    chosen = []
    for c in combos:
        if all(len(set(c) & set(o)) <= 2 for o in chosen):
            chosen.append(c)
        if len(chosen) == 40:
            break
    bits = np.zeros((len(chosen), N_BITS), dtype=np.int64)
    for i, c in enumerate(chosen):
        bits[i, list(c)] = 1
    genes = [f"gene{i:02d}" for i in range(len(chosen) - 4)] + [f"Blank-{i}" for i in range(4)]
    return bits, genes


def _psf(sig):
    z, y, x = np.indices((5, 7, 7), dtype=np.float64) - np.asarray([2, 3, 3])[:, None, None, None]
    p = np.exp(-0.5 * ((z / sig[0]) ** 2 + (y / sig[1]) ** 2 + (x / sig[2]) ** 2))
    return (p / p.sum()).astype(np.float32)


def _render(points, amps, sigma):
    vol = np.zeros(SHAPE, dtype=np.float32)
    np.add.at(vol, tuple(points.T), amps)
    return gaussian_filter(vol, sigma)


def _experiment(seed=0, zarr_root=None, write_behind=False):
    """(store, truth table, per-round shifts): 2 tiles side by side in y with a 28-px overlap.
    ``zarr_root``: keep every image on disk as OME-Zarr and load it straight into HBM (I/O fast path);
    ``write_behind``: image saves queued to the store's background writer (round 4)."""
    ds_mod = pkg("datastore")
    rng = np.random.default_rng(seed)
    bits, genes = _codebook()
    kwargs = dict(num_tiles=2, num_rounds=N_ROUNDS, num_bits=N_BITS, codebook=codebook_frame(bits, genes),
                  voxel_size_zyx_um=VOXEL)
    if zarr_root is None:
        store = ds_mod.InMemoryDataStore(**kwargs)
    else:   # images as OME-Zarr, metadata as the reference's JSON sidecars / parquet tables
        store = pkg("qi2lab_store").Qi2labDiskStore(zarr_root, device="cuda:0", num_threads=4,
                                                    write_behind=write_behind, write_threads=4, **kwargs)
    store.channel_psfs = [_psf((0.9, 1.2, 1.2)), _psf((0.9, 1.3, 1.3)), _psf((0.9, 1.4, 1.4))]
    shifts = {r: (0.0, 0.0, 0.0) if r == 0 else tuple(np.round(rng.uniform((-1, -4, -4), (1, 4, 4)), 0))
              for r in range(N_ROUNDS)}
    truth_rows, predictors = [], {}
    for t, tile in enumerate(store.tile_ids):
        origin_y = t * (SHAPE[1] - 28) * VOXEL[1]
        store.set_stage_position_zyx_um(tile, (0.0, origin_y, 0.0))
        beads = rng.integers((3, 8, 8), (SHAPE[0] - 3, SHAPE[1] - 8, SHAPE[2] - 8), size=(80, 3))
        fid = _render(beads, rng.uniform(4e4, 8e4, 80), (1.0, 1.5, 1.5)) + 100
        # coding genes only: no blank is ever planted
        n_em = 150
        code = rng.integers(0, len(genes) - 4, size=n_em)
        pos = rng.integers((3, 10, 10), (SHAPE[0] - 3, SHAPE[1] - 10, SHAPE[2] - 10), size=(n_em, 3))
        amp = rng.uniform(3e4, 6e4, n_em)
        for c, p in zip(code, pos):
            truth_rows.append((genes[c], p[0] * VOXEL[0], p[1] * VOXEL[1] + origin_y, p[2] * VOXEL[2], t))
        for r, rid in enumerate(store.round_ids):
            moved = ndi_shift(fid, shifts[r], order=1, mode="nearest")
            store.set_corrected_image(rng.poisson(moved).astype(np.uint16), tile=tile, round=rid)
        for b, bid in enumerate(store.bit_ids):
            r = b // 2
            on = bits[code, b] > 0
            img = _render(pos[on], amp[on], (1.0, 1.4, 1.4)) + 100
            img = ndi_shift(img, shifts[r], order=1, mode="nearest")
            ex = 0.561 if b % 2 == 0 else 0.635
            store.set_corrected_image(rng.poisson(img).astype(np.uint16), tile=tile, bit=bid,
                                      wavelengths_um=(ex, ex + 0.03))
            store.set_round_linker(tile, bid, r + 1)
            blob = _render(pos[on], np.ones(int(on.sum())), (0.8, 1.0, 1.0))
            blob = ndi_shift(blob / max(blob.max(), 1e-6), shifts[r], order=1, mode="nearest")
            predictors[(tile, bid)] = np.clip(blob + 0.01, 0, 1).astype(np.float32)
    truth = pd.DataFrame(truth_rows, columns=["gene_id", "global_z", "global_y", "global_x", "tile_idx"])
    return store, truth, shifts, predictors


@pytest.mark.parametrize("deformable,on_disk", [(False, False), (True, False), (True, True), (False, "write_behind")])
def test_preprocess_then_decode_recovers_the_planted_transcripts(deformable, on_disk, tmp_path):
    DR, PD, sc = pkg("DataRegistration"), pkg("PixelDecoder"), pkg("scoring")
    store, truth, shifts, predictors = _experiment(zarr_root=tmp_path / "qi2lab" if on_disk else None,
                                                   write_behind=on_disk == "write_behind")
    state = {}

    def predictor(image):
        tile, bit = state["key"]
        prob = predictors[(tile, bit)]
        flat = np.argsort(prob.ravel())[-20:]
        z, y, x = np.unravel_index(flat, prob.shape)
        return pd.DataFrame({"axis-0": z, "axis-1": y, "axis-2": x}), prob

    # the predictor callable is told which image it gets by wrapping the datastore loader
    load = store.load_local_corrected_image

    def tracking_load(tile, round=None, bit=None, return_future=True):
        if bit is not None:
            state["key"] = (store._tile(tile), store._bit(bit))
        return load(tile, round=round, bit=bit, return_future=return_future)

    store.load_local_corrected_image = tracking_load
    reg = DR.DataRegistration(store, decon_fiducial=True, decon_readout=True,
                              perform_deformable_registration=deformable, verbose=0, feature_predictor=predictor)
    reg.register_all_tiles()
    store.load_local_corrected_image = load
    if on_disk == "write_behind":   # register_all_tiles evicted its tiles and flushed the queue before returning
        assert not store._resident and not store._pending_writes
    for tile in store.tile_ids:
        for r, rid in enumerate(store.round_ids):
            T = store.load_local_round_transform_zyx_um(tile, rid)
            np.testing.assert_allclose(T[:3, 3] / np.asarray(VOXEL), shifts[r], atol=0.4, err_msg=f"{tile} {rid}")
    assert reg._is_tile_complete(store.tile_ids[0]) and reg._is_tile_complete(store.tile_ids[1])
    for tile in store.tile_ids:
        for rid in store.round_ids[1:]:
            saved = store.load_local_sofima_flow_field(tile, rid)
            assert (saved is not None) == deformable
            if deformable:   # rigid shifts only: whatever residual field is kept stays small
                assert saved[0].shape[0] == 3 and np.abs(saved[0]).max() < 2.5, np.abs(saved[0]).max()

    dec = PD.PixelDecoder(store, merfish_bits=N_BITS, verbose=0)
    dec._global_normalization_vectors(gpu_id=0)
    assert store.load_decode_normalization_vectors(None, "global")[0] is not None
    dec.decode_all_tiles(assign_to_cells=False, minimum_pixels=3, normalization_method="global",
                         target_gross_misid_rate=0.05)
    final = store.load_global_filtered_decoded_spots()
    if on_disk:   # everything the run produced is on disk in the reference's layout; a fresh handle sees it
        again = pkg("qi2lab_store").Qi2labDiskStore(tmp_path / "qi2lab")
        assert again.load_local_round_transform_zyx_um("tile0001", "round003") is not None
        assert again.load_global_filtered_decoded_spots().equals(final)
        assert again.load_decode_normalization_vectors(None, "global")[0] is not None
        assert (tmp_path / "qi2lab" / "readouts" / "tile0000" / "bit005" / "decon_data.ome.zarr" / "zarr.json").exists()
    assert len(final) > 0 and {"gene_id", "global_z", "global_y", "global_x", "distance_min", "cell_id"} <= set(final.columns)
    # planted transcripts of the overlap strip appear once in the truth per tile; de-duplicate them the same way
    truth_unique = truth.drop_duplicates(subset=["gene_id", "global_z", "global_y", "global_x"])
    res = sc.calculate_F1_with_radius(final[["global_z", "global_y", "global_x"]].to_numpy(), final["gene_id"].to_numpy(),
                                      truth_unique[["global_z", "global_y", "global_x"]].to_numpy(),
                                      truth_unique["gene_id"].to_numpy(), 0.6)[0]
    print("pipeline F1 vs planted:", res, "transcripts", len(final))
    assert res["Precision"] > 0.9, res
    assert res["Recall"] > 0.5, res
    assert not final["gene_id"].str.startswith("Blank").any() or res["Precision"] > 0.9
