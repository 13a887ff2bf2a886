"""GPU parity: chunk files <-> volumes in HBM (m3d_zarr_read_device / m3d_zarr_write_device) against the
NumPy restatement (oracle/blosc_ref.py) and the real c-blosc frames of tests/golden/blosc_frames.npz."""

import ctypes
import filecmp

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import blosc_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _volume(rng, shape, dtype):
    z, y, x = np.indices(shape)
    base = 300 + 100 * np.sin(0.2 * x) * np.cos(0.13 * y) + 10 * z
    v = rng.poisson(base).astype(np.float64)
    return (v / 977.0).astype(dtype) if np.dtype(dtype).kind == "f" else v.astype(dtype)


@pytest.fixture()
def io_handle():
    hip = pkg("_hip")
    lib = hip.lib()
    handle = ctypes.c_void_p()
    hip.check(lib.m3d_io_create(4, ctypes.byref(handle)))
    yield hip, lib, handle
    lib.m3d_io_destroy(handle)


@pytest.mark.parametrize("shape,chunks,dtype,shuffle", [
    ((5, 70, 90), (2, 32, 32), np.uint16, 2), ((20, 130, 260), (16, 64, 128), np.uint16, 2),
    ((3, 64, 64), (3, 64, 64), np.float32, 2), ((1, 33, 47), (16, 512, 512), np.uint16, 2),
    ((9, 40, 24), (4, 16, 8), np.uint8, 2), ((6, 50, 70), (4, 32, 32), np.uint16, 1),
    ((6, 50, 70), (4, 32, 32), np.float32, 0), ((4, 31, 37), (2, 9, 11), np.uint16, 2)])
def test_read_device_matches_the_restatement(tmp_path, io_handle, shape, chunks, dtype, shuffle):
    hip, lib, handle = io_handle
    vol = _volume(np.random.default_rng(4), shape, dtype)
    T = vol.dtype.itemsize
    d = tmp_path / "c"
    blosc_ref.write_array(d, vol, chunks, shuffle=shuffle)
    out = torch.full(shape, 7, dtype=getattr(torch, vol.dtype.name), device=DEV)
    stats = hip.IOStats()
    hip.check(lib.m3d_zarr_read_device(handle, str(d).encode(), hip.l3(shape), hip.l3(chunks), T, out.data_ptr(),
                                       hip.current_stream(out.device), ctypes.byref(stats)))
    assert np.array_equal(out.cpu().numpy(), vol)
    grid = [-(-s // c) for s, c in zip(shape, chunks)]
    last = [g - 1 for g in grid]
    d.joinpath(*[str(i) for i in last]).unlink()
    out.fill_(9)
    hip.check(lib.m3d_zarr_read_device(handle, str(d).encode(), hip.l3(shape), hip.l3(chunks), T, out.data_ptr(),
                                       hip.current_stream(out.device), ctypes.byref(stats)))
    want = vol.copy()
    want[tuple(slice(i * c, None) for i, c in zip(last, chunks))] = 0
    assert np.array_equal(out.cpu().numpy(), want) and stats.chunks_missing == 1


@pytest.mark.parametrize("shape,chunks,dtype", [((5, 70, 90), (2, 32, 32), np.uint16), ((20, 130, 260), (16, 64, 128), np.uint16),
                                                ((3, 64, 64), (3, 64, 64), np.float32), ((4, 31, 37), (2, 9, 11), np.uint16),
                                                ((9, 40, 24), (4, 16, 8), np.uint8)])
def test_write_device_files_equal_the_host_writer_and_decode_in_the_restatement(tmp_path, io_handle, shape, chunks, dtype):
    hip, lib, handle = io_handle
    vol = _volume(np.random.default_rng(5), shape, dtype)
    T = vol.dtype.itemsize
    dev = torch.from_numpy(vol).to(DEV)
    stats = hip.IOStats()
    d_dev, d_host = tmp_path / "dev" / "c", tmp_path / "host" / "c"
    hip.check(lib.m3d_zarr_write_device(handle, str(d_dev).encode(), hip.l3(shape), hip.l3(chunks), T, dev.data_ptr(), 5, 2, 4,
                                        hip.current_stream(dev.device), ctypes.byref(stats)))
    hip.check(lib.m3d_zarr_write_host(handle, str(d_host).encode(), hip.l3(shape), hip.l3(chunks), T, vol.ctypes.data, 5, 2, 4,
                                      ctypes.byref(stats)))
    assert np.array_equal(blosc_ref.read_array(d_dev, shape, chunks, vol.dtype), vol)
    grid = [-(-s // c) for s, c in zip(shape, chunks)]
    for z in range(grid[0]):
        for y in range(grid[1]):
            for x in range(grid[2]):
                rel = f"{z}/{y}/{x}"
                assert filecmp.cmp(d_dev / rel, d_host / rel, shallow=False), rel


def test_cblosc_frames_as_chunks(tmp_path, io_handle, golden):
    """Frames written by the real c-blosc 1.21: several blocks, a short last block that c-blosc leaves
    unshuffled, lz4 split streams, a memcpyed frame."""
    hip, lib, handle = io_handle
    g = golden("blosc_frames")
    for i, name in enumerate(g["names"]):
        raw = g[f"raw_{i}"]
        shape = (1, 1, raw.size)
        d = tmp_path / str(name) / "c" / "0" / "0"
        d.mkdir(parents=True)
        g[f"frame_{i}"].tofile(d / "0")
        out = torch.zeros(shape, dtype=getattr(torch, raw.dtype.name), device=DEV)
        hip.check(lib.m3d_zarr_read_device(handle, str(d.parents[1]).encode(), hip.l3(shape), hip.l3(shape),
                                           raw.dtype.itemsize, out.data_ptr(), hip.current_stream(out.device), None), str(name))
        assert np.array_equal(out.cpu().numpy().ravel(), raw), name


def test_frames_with_many_small_blocks(tmp_path, io_handle):
    """A chunk stored as 70 000 sixteen-byte Blosc blocks (legal, if silly): the kernels index blocks along
    one grid dimension, so the count is not limited to 65 535."""
    hip, lib, handle = io_handle
    rng = np.random.default_rng(9)
    shape = (4, 350, 400)
    vol = rng.integers(0, 4000, size=shape).astype(np.uint16)
    d = tmp_path / "c" / "0" / "0"
    d.mkdir(parents=True)
    blosc_ref.encode(vol, 2, 1, 2, blocksize=16).tofile(d / "0")
    out = torch.zeros(shape, dtype=torch.uint16, device=DEV)
    hip.check(lib.m3d_zarr_read_device(handle, str(tmp_path / "c").encode(), hip.l3(shape), hip.l3(shape), 2,
                                       out.data_ptr(), hip.current_stream(out.device), None))
    assert np.array_equal(out.cpu().numpy(), vol)


def test_image_store_round_trip_in_hbm(tmp_path):
    zio = pkg("zarr_io")
    rng = np.random.default_rng(6)
    with zio.ImageIO(num_threads=8) as io:
        vol = torch.from_numpy(_volume(rng, (40, 700, 900), np.uint16)).to(DEV)
        root = io.write_image(tmp_path / "readouts" / "tile0000" / "bit001" / "decon_data", vol,
                              ome_scale=(0.315, 0.098, 0.098), extra_attributes={"deconvolution": True})
        w = dict(io.last_stats)
        back = io.read_image(root, device=DEV)
        assert back.dtype == torch.uint16 and torch.equal(back, vol)
        assert np.array_equal(io.read_image(root), vol.cpu().numpy())           # all-host path, same files
        assert w["chunks"] == 3 * 2 * 2 and w["file_bytes"] < w["raw_bytes"]
        flow = torch.from_numpy(rng.normal(size=(3, 5, 30, 40)).astype(np.float32)).to(DEV)
        io.write_image(tmp_path / "flow", flow)
        assert torch.equal(io.read_image(tmp_path / "flow", device=DEV), flow)
        pred = torch.from_numpy(rng.uniform(size=(20, 300, 520)).astype(np.float32)).to(DEV)
        io.write_image(tmp_path / "feature_predictor_data", pred)
        assert torch.equal(io.read_image(tmp_path / "feature_predictor_data", device=DEV), pred)


def test_tile_read_ahead_hands_over_the_same_images(tmp_path):
    """prefetch_tile loads on a side stream in a background thread; load_* must return the same tensors as a
    direct read, consume the read-ahead entry, and a later write must not be shadowed by a stale entry."""
    ds = pkg("datastore")
    from helpers import codebook_frame

    rng = np.random.default_rng(8)
    store = ds.ZarrImageDataStore(tmp_path / "qi2lab", device=DEV, num_threads=4, num_tiles=2, num_rounds=1,
                                  num_bits=3, codebook=codebook_frame(np.ones((3, 3), dtype=np.int64)))
    imgs = {}
    for t in range(2):
        for b in range(3):
            img = rng.integers(90, 5000, size=(20, 140, 300)).astype(np.uint16)
            pred = rng.uniform(size=img.shape).astype(np.float32)
            store.set_corrected_image(img, tile=t, bit=b)
            store.save_local_feature_predictor_image(pred, tile=t, bit=b)
            imgs[(t, b)] = (img, pred)
    store.save_local_deconvolved_readout_image(imgs[(1, 2)][0] // 2, tile=1, bit=2)   # decon shadows corrected
    assert store.prefetch_tile(1) == 6 and len(store._prefetched) == 6
    assert store.prefetch_tile(1) == 0                                               # already queued
    busy = torch.ones((2048, 2048), device=DEV)
    for _ in range(20):
        busy = busy @ busy * 1e-4                                                      # compute on the main stream meanwhile
    for b in range(3):
        want = imgs[(1, b)][0] // 2 if b == 2 else imgs[(1, b)][0]
        got = store.load_local_readout_image(1, bit=b, return_future=False)
        assert got.is_cuda and np.array_equal(got.cpu().numpy(), want)
        assert np.array_equal(store.load_local_feature_predictor_image(1, bit=b, return_future=False).cpu().numpy(),
                              imgs[(1, b)][1])
    assert not store._prefetched
    store.prefetch_tile(0, bits=store.bit_ids[:1])
    store.set_corrected_image(imgs[(0, 0)][0] + 7, tile=0, bit=0)                     # invalidates the read-ahead
    assert np.array_equal(store.load_local_readout_image(0, bit=0, return_future=False).cpu().numpy(), imgs[(0, 0)][0] + 7)
    torch.cuda.synchronize()


def test_decode_worker_reopens_the_datastore_by_path(tmp_path):
    """decode_tiles_worker (the per-GPU child of decode_all_tiles, PixelDecoder.py:207-309) gets a path, not an
    object: it must find images, normalisation vectors and linkers on disk and leave its tables there."""
    PD, qs = pkg("PixelDecoder"), pkg("qi2lab_store")
    from helpers import codebook_frame

    rng = np.random.default_rng(12)
    bits = np.array([[1, 1, 1, 1, 0, 0], [0, 0, 1, 1, 1, 1], [1, 1, 0, 0, 1, 1]], dtype=np.int64)
    store = qs.Qi2labDiskStore(tmp_path / "ds", device=DEV, num_tiles=2, num_rounds=1, num_bits=6,
                               codebook=codebook_frame(bits, ["g0", "g1", "Blank-0"]))
    shape = (6, 64, 64)
    for t in range(2):
        store.set_stage_position_zyx_um(t, (0.0, 10.0 * t, 0.0))
        pos = rng.integers((1, 6, 6), (5, 58, 58), size=(12, 3))
        code = rng.integers(0, 2, size=12)
        for b in range(6):
            vol = rng.normal(100, 3, size=shape)
            for (z, y, x), c in zip(pos, code):
                if bits[c, b]:
                    vol[z - 1:z + 2, y - 2:y + 3, x - 2:x + 3] += 900
            store.set_corrected_image(np.clip(vol, 0, 65535).astype(np.uint16), tile=t, bit=b)
            store.save_local_feature_predictor_image(np.ones(shape, dtype=np.float32), tile=t, bit=b)
            store.set_round_linker(t, b, 1)
        store.save_local_round_transform_zyx_um(np.eye(4, dtype=np.float32), tile=t, round=0)
    for kind in ("global", "iterative"):   # the worker loads both, as the reference's does (:272-277)
        store.save_decode_normalization_vectors(None, kind, np.full(6, 900.0, np.float32), np.full(6, 100.0, np.float32))
    PD.decode_tiles_worker(tmp_path / "ds", [0, 1], 0, 6, 0, "auto", (3, 1, 1), (0.5, 10.0), 3, 0.1, "global")
    fresh = qs.Qi2labDiskStore(tmp_path / "ds")
    for t in range(2):
        table = fresh.load_local_decoded_spots(tile=t)
        assert table is not None and len(table) >= 8 and set(table["gene_id"]) <= {"g0", "g1", "Blank-0"}
        assert (table["tile_idx"] == t).all()


def test_write_behind_leaves_the_same_bytes_on_disk_and_keeps_the_tile_in_hbm(tmp_path):
    """``write_behind=True``: saves return at once and run on a background thread with its own stream; what lands on
    disk is byte-identical to the synchronous store's files, a load of a volume whose write is still queued comes from
    HBM, ``evict_tile`` + ``flush`` leave only the disk copy, and ``prefetch_tile_inputs`` reads the preprocess's
    inputs ahead in the order the operators ask for them."""
    ds = pkg("datastore")
    from helpers import codebook_frame

    rng = np.random.default_rng(21)
    cb = codebook_frame(np.ones((3, 3), dtype=np.int64))
    kw = dict(device=DEV, num_threads=2, num_tiles=2, num_rounds=2, num_bits=3, codebook=cb)
    lazy = ds.ZarrImageDataStore(tmp_path / "lazy", write_behind=True, prefetch_threads=2, write_threads=3,
                                 max_pending_writes=2, **kw)
    eager = ds.ZarrImageDataStore(tmp_path / "eager", **kw)
    vols = {}
    for t in range(2):
        for r in range(2):
            v = torch.from_numpy(rng.integers(80, 9000, size=(18, 130, 520)).astype(np.uint16)).to(DEV)
            vols[(t, "round", r)] = v
            for s in (lazy, eager):
                s.set_corrected_image(v, tile=t, round=r)
        for b in range(3):
            v = torch.from_numpy(rng.integers(80, 9000, size=(18, 130, 520)).astype(np.uint16)).to(DEV)
            p = torch.from_numpy(rng.uniform(size=(18, 130, 520)).astype(np.float32)).to(DEV)
            vols[(t, "bit", b)] = (v, p)
            for s in (lazy, eager):
                s.set_corrected_image(v, tile=t, bit=b)
                s.save_local_feature_predictor_image(p, tile=t, bit=b)
    assert len(lazy._pending_writes) <= 2                           # back-pressure held the queue at its bound
    got = lazy.load_local_corrected_image(1, bit=2, return_future=False)
    assert got.data_ptr() == vols[(1, "bit", 2)][0].data_ptr()      # still resident: no I/O
    lazy.flush()
    assert not lazy._pending_writes
    for path in sorted((tmp_path / "eager").rglob("*")):
        if path.is_file():
            twin = tmp_path / "lazy" / path.relative_to(tmp_path / "eager")
            assert twin.exists() and twin.read_bytes() == path.read_bytes(), path
    assert lazy.evict_tile(1) == 2 + 3 + 3 and lazy.evict_tile(0) == 8 and not lazy._resident
    # read-ahead of a tile's preprocess inputs: fiducials of both rounds, then readout + predictor per bit
    assert lazy.prefetch_tile_inputs(1) == 2 + 6
    busy = torch.ones((1024, 1024), device=DEV)
    for _ in range(10):
        busy = busy @ busy * 1e-3
    for r in range(2):
        assert torch.equal(lazy.load_local_corrected_image(1, round=r, return_future=False), vols[(1, "round", r)])
    for b in range(3):
        assert torch.equal(lazy.load_local_corrected_image(1, bit=b, return_future=False), vols[(1, "bit", b)][0])
        assert torch.equal(lazy.load_local_feature_predictor_image(1, bit=b, return_future=False), vols[(1, "bit", b)][1])
    assert not lazy._prefetched
    # a deconvolved volume saved behind the compute shadows the corrected one at once and survives eviction on disk
    half = (vols[(0, "bit", 1)][0].to(torch.int32) // 2).to(torch.uint16)
    lazy.save_local_deconvolved_readout_image(half, tile=0, bit=1)
    assert lazy.load_local_readout_image(0, bit=1, return_future=False).data_ptr() == half.data_ptr()
    assert lazy._image_shape(0, bit=1, kind="decon_data") == (18, 130, 520)
    lazy.evict_tile(0)
    assert torch.equal(lazy.load_local_readout_image(0, bit=1, return_future=False), half)   # waits for the write
    torch.cuda.synchronize()
