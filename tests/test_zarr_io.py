"""CPU: the datastore image I/O path (include/m3d_io.h, zarr_io.py) - Blosc 1 frames against the real
c-blosc's (tests/golden/blosc_frames.npz), the Zarr v3 chunk layout against the NumPy restatement in
oracle/blosc_ref.py, and the OME-Zarr metadata the reference's writer produces."""

import ctypes
import json

import numpy as np
import pytest

from conftest import pkg
from oracle import blosc_ref


def _cases(golden):
    g = golden("blosc_frames")
    return g, [(i, str(n)) for i, n in enumerate(g["names"])]


def test_oracle_decodes_every_cblosc_frame(golden):
    g, cases = _cases(golden)
    for i, name in cases:
        raw = g[f"raw_{i}"]
        assert np.array_equal(blosc_ref.decode(g[f"frame_{i}"]), raw.view(np.uint8)), name
        assert np.array_equal(blosc_ref.decode(g[f"ours_{i}"]), raw.view(np.uint8)), name


def test_host_codec_decodes_cblosc_frames_and_writes_frames_cblosc_accepted(golden):
    """``ours_i`` are frames of m3d_blosc_encode_host that c-blosc's blosc_decompress returned the input
    from (make_golden_blosc.py); the encoder must still produce exactly those bytes."""
    hip = pkg("_hip")
    lib = hip.lib()
    g, cases = _cases(golden)
    for i, name in cases:
        raw = np.ascontiguousarray(g[f"raw_{i}"])
        for key in ("frame", "ours"):
            frame = np.ascontiguousarray(g[f"{key}_{i}"])
            head = hip.BloscHeader()
            assert lib.m3d_blosc_info(frame.ctypes.data, frame.size, ctypes.byref(head)) == 0
            want = blosc_ref.parse_header(frame)
            assert (head.nbytes, head.blocksize, head.cbytes, head.typesize, head.shuffle, head.codec) == \
                (want["nbytes"], want["blocksize"], want["cbytes"], want["typesize"], want["shuffle"], want["codec"])
            assert bool(head.memcpyed) == want["memcpyed"] and bool(head.split) == want["split"]
            out = np.zeros_like(raw)
            hip.check(lib.m3d_blosc_decode_host(frame.ctypes.data, frame.size, out.ctypes.data, out.nbytes), name)
            assert np.array_equal(out, raw), (name, key)
        want = blosc_ref.parse_header(g[f"frame_{i}"])
        blocksize = blosc_ref.parse_header(g[f"ours_{i}"])["blocksize"]
        cap = lib.m3d_blosc_bound(raw.nbytes, blocksize)
        buf = np.zeros(cap, dtype=np.uint8)
        got = ctypes.c_size_t(0)
        hip.check(lib.m3d_blosc_encode_host(raw.ctypes.data, raw.nbytes, raw.dtype.itemsize, 5, want["shuffle"],
                                            want["codec"], blocksize, buf.ctypes.data, cap, ctypes.byref(got)), name)
        assert np.array_equal(buf[: got.value], g[f"ours_{i}"]), name
        if want["codec"] == 4 and not want["memcpyed"] and "random" not in name:
            assert np.array_equal(buf[: got.value], g[f"frame_{i}"]), f"{name}: differs from c-blosc's own bytes"


def test_truncated_and_foreign_frames_are_rejected(golden):
    hip = pkg("_hip")
    lib = hip.lib()
    g, _ = _cases(golden)
    frame = np.ascontiguousarray(g["frame_1"])
    raw = g["raw_1"]
    out = np.zeros_like(raw)
    with pytest.raises(ValueError):
        hip.check(lib.m3d_blosc_decode_host(frame.ctypes.data, frame.size - 9, out.ctypes.data, out.nbytes))
    with pytest.raises(ValueError):
        hip.check(lib.m3d_blosc_decode_host(frame.ctypes.data, frame.size, out.ctypes.data, out.nbytes - 2))
    bad = frame.copy()
    bad[0] = 3
    with pytest.raises(ValueError):
        hip.check(lib.m3d_blosc_decode_host(bad.ctypes.data, bad.size, out.ctypes.data, out.nbytes))
    bad = frame.copy()
    bad[40:60] ^= 0x5A  # damaged zstd stream
    with pytest.raises(ValueError):
        hip.check(lib.m3d_blosc_decode_host(bad.ctypes.data, bad.size, out.ctypes.data, out.nbytes))


def _volume(rng, shape, dtype):
    z, y, x = np.indices(shape)
    base = 300 + 100 * np.sin(0.2 * x) * np.cos(0.13 * y) + 10 * z
    v = rng.poisson(base).astype(np.float64)
    return (v / 977.0).astype(dtype) if np.dtype(dtype).kind == "f" else v.astype(dtype)


@pytest.mark.parametrize("shape,chunks,dtype", [((5, 70, 90), (2, 32, 32), np.uint16), ((3, 64, 64), (3, 64, 64), np.float32),
                                                ((1, 33, 47), (16, 512, 512), np.uint16), ((9, 40, 24), (4, 16, 8), np.uint8)])
def test_chunk_files_round_trip_against_the_restatement(tmp_path, shape, chunks, dtype):
    hip = pkg("_hip")
    lib = hip.lib()
    handle = ctypes.c_void_p()
    hip.check(lib.m3d_io_create(3, ctypes.byref(handle)))
    try:
        vol = _volume(np.random.default_rng(1), shape, dtype)
        T = vol.dtype.itemsize
        stats = hip.IOStats()
        # product writes, restatement reads
        d1 = tmp_path / "ours" / "c"
        hip.check(lib.m3d_zarr_write_host(handle, str(d1).encode(), hip.l3(shape), hip.l3(chunks), T, vol.ctypes.data,
                                          5, 2, 4, ctypes.byref(stats)))
        assert stats.raw_bytes == vol.nbytes and stats.chunks == np.prod([-(-s // c) for s, c in zip(shape, chunks)])
        assert np.array_equal(blosc_ref.read_array(d1, shape, chunks, vol.dtype), vol)
        # every chunk file holds the full chunk shape (edge chunks zero padded)
        first = np.fromfile(d1 / "0" / "0" / "0", dtype=np.uint8)
        assert blosc_ref.parse_header(first)["nbytes"] == int(np.prod(chunks)) * T
        # restatement writes, product reads; one chunk removed reads as the fill value
        d2 = tmp_path / "ref" / "c"
        blosc_ref.write_array(d2, vol, chunks)
        out = np.full(shape, 7, dtype=vol.dtype)
        hip.check(lib.m3d_zarr_read_host(handle, str(d2).encode(), hip.l3(shape), hip.l3(chunks), T, out.ctypes.data,
                                         ctypes.byref(stats)))
        assert np.array_equal(out, vol) and stats.chunks_missing == 0
        grid = [-(-s // c) for s, c in zip(shape, chunks)]
        last = [g - 1 for g in grid]
        (d2.joinpath(*[str(i) for i in last])).unlink()
        hip.check(lib.m3d_zarr_read_host(handle, str(d2).encode(), hip.l3(shape), hip.l3(chunks), T, out.ctypes.data,
                                         ctypes.byref(stats)))
        want = vol.copy()
        want[tuple(slice(i * c, None) for i, c in zip(last, chunks))] = 0
        assert np.array_equal(out, want) and stats.chunks_missing == 1
        # a frame of the wrong size is an error, not garbage
        blosc_ref.encode(np.zeros(10, dtype=vol.dtype), T).tofile(d2 / "0" / "0" / "0")
        with pytest.raises(ValueError):
            hip.check(lib.m3d_zarr_read_host(handle, str(d2).encode(), hip.l3(shape), hip.l3(chunks), T,
                                             out.ctypes.data, ctypes.byref(stats)))
    finally:
        lib.m3d_io_destroy(handle)


def test_default_chunks_and_store_names_follow_the_reference_rules():
    zio = pkg("zarr_io")
    assert zio.default_chunks((64, 2048, 2048)) == [16, 512, 512]
    assert zio.default_chunks((8, 300, 2048)) == [8, 300, 512]
    assert zio.default_chunks((3, 11, 127, 127)) == [1, 11, 127, 127]
    assert zio.default_chunks((2, 3, 20, 600, 600)) == [1, 1, 16, 512, 512]
    assert zio.default_chunks((700, 100)) == [512, 100]
    assert zio.image_store_path("/a/b/decon_data").name == "decon_data.ome.zarr"
    assert zio.image_store_path("/a/b/decon_data.ome.zarr").name == "decon_data.ome.zarr"
    with pytest.raises(ValueError):
        zio.image_store_path("/a/b/decon_data.zarr")
    with pytest.raises(ValueError):
        zio.image_store_path("/a/b/decon_data.tif")
    assert zio.normalize_transform((0.3, 0.1, 0.1), 4, 1.0) == [1.0, 0.3, 0.1, 0.1]
    assert zio.normalize_transform(None, 3, 0.0) == [0.0, 0.0, 0.0]
    assert zio.normalize_transform((1, 2), 3, 1.0) == [1.0, 1.0, 1.0]


def test_ome_zarr_store_on_the_host_path(tmp_path):
    zio = pkg("zarr_io")
    rng = np.random.default_rng(2)
    with zio.ImageIO(num_threads=2) as io:
        vol = _volume(rng, (20, 100, 130), np.uint16)
        root = io.write_image(tmp_path / "tile0000" / "round001" / "corrected_data", vol, chunks=(8, 64, 64),
                              ome_scale=(0.315, 0.098, 0.098), ome_translation=(1.0, 2.0, 3.0),
                              extra_attributes={"psf_idx": 1, "gain_correction": True,
                                                "stage_zyx_um": np.array([1.5, 2.5, 3.5])})
        assert root.name == "corrected_data.ome.zarr"
        group = json.loads((root / "zarr.json").read_text())
        arr = json.loads((root / "0" / "zarr.json").read_text())
        assert group["zarr_format"] == 3 and group["node_type"] == "group"
        ms = group["attributes"]["ome"]["multiscales"][0]
        assert group["attributes"]["ome"]["version"] == "0.5"
        assert [a["name"] for a in ms["axes"]] == ["z", "y", "x"] and ms["axes"][0]["unit"] == "micrometer"
        tr = ms["datasets"][0]["coordinateTransformations"]
        assert ms["datasets"][0]["path"] == "0" and tr[0] == {"type": "scale", "scale": [0.315, 0.098, 0.098]}
        assert tr[1] == {"type": "translation", "translation": [1.0, 2.0, 3.0]}
        assert arr["data_type"] == "uint16" and arr["shape"] == [20, 100, 130] and arr["fill_value"] == 0
        assert arr["chunk_grid"] == {"name": "regular", "configuration": {"chunk_shape": [8, 64, 64]}}
        assert arr["chunk_key_encoding"]["name"] == "default" and arr["dimension_names"] == ["z", "y", "x"]
        assert arr["codecs"][0] == {"name": "bytes", "configuration": {"endian": "little"}}
        blosc = arr["codecs"][1]
        assert blosc["name"] == "blosc" and blosc["configuration"] == {
            "cname": "zstd", "clevel": 5, "shuffle": "bitshuffle", "typesize": 2, "blocksize": 0}
        assert (root / "0" / "c" / "2" / "1" / "2").exists()
        assert io.image_shape(root) == (20, 100, 130)
        assert np.array_equal(io.read_image(tmp_path / "tile0000" / "round001" / "corrected_data"), vol)
        assert io.last_stats["raw_bytes"] == vol.nbytes and io.last_stats["chunks"] == 3 * 2 * 3
        assert np.array_equal(blosc_ref.read_array(root / "0" / "c", vol.shape, (8, 64, 64), vol.dtype), vol)
        assert zio.read_extra_attributes(root) == {"psf_idx": 1, "gain_correction": True, "stage_zyx_um": [1.5, 2.5, 3.5]}
        zio.write_extra_attributes(root, {"deconvolution": True})
        assert zio.read_extra_attributes(root)["deconvolution"] is True and zio.read_extra_attributes(root)["psf_idx"] == 1
        # overwrite with default chunks; float64 is stored as float32 (:2302-2303)
        io.write_image(root, vol[:3].astype(np.float64))
        back = io.read_image(root)
        assert back.dtype == np.float32 and np.array_equal(back, vol[:3].astype(np.float32))
        assert json.loads((root / "0" / "zarr.json").read_text())["chunk_grid"]["configuration"]["chunk_shape"] == [3, 100, 130]
        # 4-D flow field (3, z, y, x) and a 2-D projection
        flow = rng.normal(size=(3, 5, 30, 40)).astype(np.float32)
        io.write_image(tmp_path / "opticalflow_xform_px", flow, extra_attributes={"sofima_status": "ok"})
        assert np.array_equal(io.read_image(tmp_path / "opticalflow_xform_px"), flow)
        meta = json.loads((tmp_path / "opticalflow_xform_px.ome.zarr" / "0" / "zarr.json").read_text())
        assert meta["chunk_grid"]["configuration"]["chunk_shape"] == [1, 5, 30, 40] and meta["dimension_names"] == ["c", "z", "y", "x"]
        proj = vol.max(axis=0)
        io.write_image(tmp_path / "proj", proj)
        assert np.array_equal(io.read_image(tmp_path / "proj"), proj)
        with pytest.raises(FileNotFoundError):
            io.read_image(tmp_path / "missing")
        with pytest.raises(ValueError):
            io.write_image(tmp_path / "bad", np.zeros(4, dtype=np.uint16))


def test_disk_backed_datastore_keeps_the_reference_layout(tmp_path):
    ds = pkg("datastore")
    from helpers import codebook_frame

    bits = np.eye(4, dtype=np.int64)[:, :4]
    store = ds.ZarrImageDataStore(tmp_path / "qi2lab", num_tiles=2, num_rounds=2, num_bits=4,
                                  codebook=codebook_frame(np.ones((3, 4), dtype=np.int64)), num_threads=2)
    rng = np.random.default_rng(0)
    img = rng.integers(90, 400, size=(6, 40, 50)).astype(np.uint16)
    store.set_corrected_image(img, tile=1, round=0)
    store.set_corrected_image(img + 1, tile="tile0001", bit="bit003", wavelengths_um=(0.635, 0.67))
    root = tmp_path / "qi2lab"
    assert (root / "fiducial" / "tile0001" / "round001" / "corrected_data.ome.zarr" / "0" / "c" / "0" / "0" / "0").exists()
    assert (root / "readouts" / "tile0001" / "bit003" / "corrected_data.ome.zarr" / "zarr.json").exists()
    assert np.array_equal(store.load_local_corrected_image(1, round=0, return_future=False), img)
    assert np.array_equal(store.load_local_corrected_image(1, bit=2).result(), img + 1)
    assert store.load_local_corrected_image(0, round=0) is None
    assert store._image_shape(1, round=0) == img.shape and store._image_shape(0, round=0) is None
    assert store._image_shape(root / "fiducial" / "tile0001" / "round001" / "corrected_data") == img.shape
    assert store._image_shape(root / "fiducial" / "tile0000" / "round001" / "corrected_data") is None
    # deconvolved images shadow the corrected ones (qi2labDataStore.py:4709)
    assert np.array_equal(store.load_local_readout_image(1, bit=2, return_future=False), img + 1)
    store.save_local_deconvolved_readout_image(img * 2, tile=1, bit=2)
    assert (root / "readouts" / "tile0001" / "bit003" / "decon_data.ome.zarr").exists()
    assert np.array_equal(store.load_local_readout_image(1, bit=2, return_future=False), img * 2)
    assert store._image_shape(1, bit=2, kind="decon_data") == img.shape
    store.save_local_deconvolved_fiducial_image(img * 3, tile=1, round=0)
    assert np.array_equal(store.load_local_fiducial_image(1, round=0, return_future=False), img * 3)
    pred = rng.uniform(size=img.shape).astype(np.float32)
    store.save_local_feature_predictor_image(pred, tile=1, bit=2)
    assert (root / "readouts" / "tile0001" / "bit003" / f"{store.feature_predictor_folder_name}_data.ome.zarr").exists()
    assert np.array_equal(store.load_local_feature_predictor_image(1, bit=2, return_future=False), pred)
    flow = rng.normal(size=(3, 2, 5, 6)).astype(np.float32)
    store.save_local_sofima_flow_field(flow, tile=1, round=1, map_stride_zyx_px=[4, 16, 16], sofima_status="ok")
    got, attrs = store.load_local_sofima_flow_field(1, 1)
    assert np.array_equal(got, flow) and attrs == {"map_stride_zyx_px": [4, 16, 16], "sofima_status": "ok"}
    assert (root / "fiducial" / "tile0001" / "round002" / "opticalflow_xform_px.ome.zarr").exists()
    assert store.load_local_sofima_flow_field(0, 1) is None
    # predictor localisations as parquet; directory creation is idempotent
    # (reference tests/test_datastore_feature_predictor.py:10-33)
    import pandas as pd

    spots = pd.DataFrame({"z": [1], "y": [2], "x": [3]})
    assert not store.has_feature_predictor_spots(1, 2)
    store.save_local_feature_predictor_spots(spots, tile="tile0001", bit="bit003")
    store.save_local_feature_predictor_spots(spots, tile="tile0001", bit="bit003")
    path = root / f"{store.feature_predictor_folder_name}_localizations" / "tile0001" / "bit003.parquet"
    assert path.exists() and store.has_feature_predictor_spots(1, 2)
    assert store.load_local_feature_predictor_spots(1, 2).equals(spots)


def test_codec_round_trips_property():
    """Randomised sizes, element widths, shuffles, block sizes and contents: the C codec and the NumPy
    restatement decode each other's frames (hypothesis)."""
    from hypothesis import given, settings, strategies as st

    hip = pkg("_hip")
    lib = hip.lib()

    @settings(max_examples=120, deadline=None)
    @given(st.data())
    def run(data):
        T = data.draw(st.sampled_from([1, 2, 4, 8]))
        n_el = data.draw(st.integers(min_value=1, max_value=5000))
        shuffle = data.draw(st.sampled_from([0, 1, 2]))
        codec = data.draw(st.sampled_from([4, 1]))
        kind = data.draw(st.sampled_from(["random", "smooth", "zeros"]))
        blocks = data.draw(st.sampled_from([0, 1, 2, 3, 7]))
        rng = np.random.default_rng(data.draw(st.integers(0, 2**31)))
        if kind == "random":
            raw = rng.integers(0, 256, size=n_el * T, dtype=np.uint8)
        elif kind == "zeros":
            raw = np.zeros(n_el * T, dtype=np.uint8)
        else:
            raw = (np.cumsum(rng.integers(-2, 3, size=n_el)) + 1000).astype(f"<u{T}" if T > 1 else "u1").view(np.uint8)
        nbytes = raw.size
        blocksize = 0 if blocks == 0 else max(T, (nbytes // blocks) // T * T)
        cap = lib.m3d_blosc_bound(nbytes, blocksize)
        buf = np.zeros(cap, dtype=np.uint8)
        got = ctypes.c_size_t(0)
        hip.check(lib.m3d_blosc_encode_host(raw.ctypes.data, nbytes, T, 5, shuffle, codec, blocksize, buf.ctypes.data, cap,
                                            ctypes.byref(got)))
        frame = buf[: got.value].copy()
        assert got.value <= cap
        assert np.array_equal(blosc_ref.decode(frame), raw)
        out = np.empty_like(raw)
        hip.check(lib.m3d_blosc_decode_host(frame.ctypes.data, frame.size, out.ctypes.data, out.size))
        assert np.array_equal(out, raw)
        if codec == 4:
            ref_frame = np.ascontiguousarray(blosc_ref.encode(raw, T, 5, shuffle, blocksize or 256 * 1024))
            out[:] = 0
            hip.check(lib.m3d_blosc_decode_host(ref_frame.ctypes.data, ref_frame.size, out.ctypes.data, out.size))
            assert np.array_equal(out, raw)
            assert np.array_equal(ref_frame, frame)          # same container bytes from both writers

    run()


def test_write_behind_without_a_device_stays_synchronous_and_lazy_fallbacks_do_not_read(tmp_path, monkeypatch):
    """Round 4.  (a) ``write_behind=True`` needs a device (the queue holds HBM tensors); without one the store writes
    synchronously as before - nothing pending, nothing resident, files on disk when ``save`` returns.  (b) The
    deconvolved-else-corrected fallback of ``load_local_readout_image`` / ``load_local_fiducial_image`` must not READ
    the corrected image when a deconvolved one exists (round 3 evaluated ``get(k, fallback)`` eagerly: on a disk
    store every decode of a bit read 0.5 GB it did not need)."""
    from pathlib import Path

    from helpers import codebook_frame

    ds = pkg("datastore")
    store = ds.ZarrImageDataStore(tmp_path / "qi2lab", device=None, write_behind=True, num_tiles=1, num_rounds=1,
                                  num_bits=2, codebook=codebook_frame(np.ones((2, 2), dtype=np.int64)))
    assert not store._write_behind
    img = np.arange(4 * 20 * 24, dtype=np.uint16).reshape(4, 20, 24)
    store.set_corrected_image(img, tile=0, bit=0)
    store.set_corrected_image(img + 1, tile=0, round=0)
    store.save_local_deconvolved_readout_image(img * 2, tile=0, bit=0)
    store.save_local_deconvolved_fiducial_image(img * 3, tile=0, round=0)
    assert not store._pending_writes and not store._resident
    assert (tmp_path / "qi2lab" / "readouts" / "tile0000" / "bit001" / "decon_data.ome.zarr" / "0" / "zarr.json").exists()
    store.flush()                                                   # a no-op here, harmless
    assert store.evict_tile(0) == 0
    reads = []
    real = store._io.read_image
    monkeypatch.setattr(store._io, "read_image", lambda path, device=None: (reads.append(Path(path).name), real(path, device=device))[1])
    assert np.array_equal(store.load_local_readout_image(0, bit=0, return_future=False), img * 2)
    assert np.array_equal(store.load_local_fiducial_image(0, round=0, return_future=False), img * 3)
    assert reads == ["decon_data", "decon_data"], reads              # the corrected images were not touched
    assert store.load_local_readout_image(0, bit=1, return_future=False) is None   # nothing stored for that bit
