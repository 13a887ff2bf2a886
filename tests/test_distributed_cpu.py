"""CPU, gloo, world_size 2: what the N > 1 path adds on the host - tile / round / bit shares per
rank (PixelDecoder.py:4811-4820, DataRegistration.py:2137-2141), the pooled order statistics
whose radix histograms are summed over ranks (PixelDecoder.py:1113-1177) and bench.py's
max-over-ranks timing.  The per-rank device kernels are covered by the -m gpu tests; here the
local histogram is emulated in NumPy so the collective logic runs without a GPU."""

import os
import socket
import sys
import traceback
from pathlib import Path

import numpy as np
import pandas as pd
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _float_keys(values: np.ndarray) -> np.ndarray:
    """Monotone uint32 keys of float32 values (the inverse of kernels._key_to_float)."""
    bits = values.astype(np.float32).view(np.uint32)
    return np.where(bits & 0x80000000, ~bits, bits | 0x80000000).astype(np.uint32)


def _numpy_local_hist(self, pass_idx, prefix, device):
    hist = np.zeros(256, dtype=np.int64)
    for values, transform, sub, flt, thr in self.parts:
        v = values.numpy().astype(np.float32)
        if transform:
            v = np.maximum(v - np.float32(sub), np.float32(0))
        if flt == 1:
            v = v[v < np.float32(thr)]
        elif flt == 2:
            v = v[v > np.float32(thr)]
        k = _float_keys(v)
        shift = 24 - 8 * pass_idx
        if pass_idx:
            k = k[(k >> np.uint32(shift + 8)) == np.uint32(prefix >> (shift + 8))]
        hist += np.bincount((k >> np.uint32(shift)) & np.uint32(0xFF), minlength=256)
    return torch.from_numpy(hist)


def _worker(rank, world, port, ret):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        sys.path.insert(0, str(ROOT))
        sys.path.insert(0, str(ROOT / "tests"))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from conftest import pkg

        kernels, PD, DR, DS = pkg("kernels"), pkg("PixelDecoder"), pkg("DataRegistration"), pkg("datastore")
        out = {"dist_ready": kernels._dist_ready()}

        # shares: every rank gets one contiguous chunk, together they cover the list once
        cb = pd.DataFrame({"gene_id": ["a", "b"], **{f"bit{i + 1:02d}": [1, 0] for i in range(16)}})
        store = DS.InMemoryDataStore(num_tiles=5, num_rounds=8, num_bits=16, codebook=cb)
        store.channel_psfs = [np.ones((3, 3, 3), dtype=np.float32)] * 3
        reg = DR.DataRegistration(store, num_gpus=8, verbose=0)
        shares, distributed = reg._shares(list(store.round_ids[1:]))
        out["rounds"] = (shares, distributed)
        out["bits"] = reg._shares(list(store.bit_ids))
        out["tiles_dr"] = DR.DataRegistration._tiles_for_this_rank(list(store.tile_ids))
        out["tiles_dr_few"] = DR.DataRegistration._tiles_for_this_rank(["only"])
        reg._tile_parallel = True          # a rank that owns whole tiles keeps every round / bit of them
        out["bits_tile_parallel"] = reg._shares(list(store.bit_ids))
        reg._tile_parallel = False
        dec = PD.PixelDecoder.__new__(PD.PixelDecoder)
        dec._num_gpus = 4
        out["tiles"] = dec._shard(list(range(5)))

        # pooled order statistics over both ranks
        rng = np.random.default_rng(100 + rank)
        a = rng.normal(5, 3, size=4001 + 17 * rank).astype(np.float32)
        b = rng.gamma(2.0, 2.0, size=2500).astype(np.float32)
        kernels.PooledValues._local_hist = _numpy_local_hist
        pool = kernels.PooledValues()
        pool.add(torch.from_numpy(a), sub=1.5)
        pool.add(torch.from_numpy(b), below=6.0)
        out["count"] = pool.count()
        # kth / percentile / median use parts[0][0].device -> CPU tensors here
        out["p999"] = pool.percentile(99.9)
        out["p50"] = pool.percentile(50.0)
        out["median"] = pool.median()
        out["local"] = (np.maximum(a - np.float32(1.5), 0), b[b < 6.0])
        # a local-only pool (group=False) ignores the other rank
        solo = kernels.PooledValues(group=False)
        solo.add(torch.from_numpy(a))
        out["solo_median"] = solo.median()
        out["solo_ref"] = float(np.median(a))

        # bench timing rule
        sys.argv = ["bench.py"]
        import bench

        out["agg"] = bench.aggregate_over_ranks(1.0 + rank, 10.0 * (rank + 1), world, torch.device("cpu"))
        # the ranks' card states side by side (rank 1's card the slower, hotter one); a rank without a record is left out
        state = {"sclk_mhz": {"mean": 2140.0 - 60.0 * rank}, "power_w": {"mean": 1350.0 + 8.0 * rank},
                 "hbm_c": {"mean": 66.0 + 5.0 * rank}}
        out["cards"] = bench.card_state_over_ranks(state, world, torch.device("cpu"))
        out["cards_one_missing"] = bench.card_state_over_ranks(state if rank == 0 else None, world, torch.device("cpu"))
        out["cards_none"] = bench.card_state_over_ranks(None, world, torch.device("cpu"))
        dist.barrier()
        dist.destroy_process_group()
        ret[rank] = out
    except Exception:
        ret[rank] = {"error": traceback.format_exc()}
        raise


def test_two_rank_shares_pooled_statistics_and_timing():
    world = 2
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=240)
        res = {k: dict(v) for k, v in ret.items()}
    assert len(res) == world, res
    for r in range(world):
        assert "error" not in res[r], res[r].get("error")
        assert res[r]["dist_ready"]
    rounds = [f"round{i + 1:03d}" for i in range(1, 8)]
    assert res[0]["rounds"] == ([rounds[:4]], True) and res[1]["rounds"] == ([rounds[4:]], True)
    bits = [f"bit{i + 1:03d}" for i in range(16)]
    assert res[0]["bits"] == ([bits[:8]], True) and res[1]["bits"] == ([bits[8:]], True)
    assert res[0]["tiles"] == ([[0, 1, 2]], True) and res[1]["tiles"] == ([[3, 4]], True)
    tiles = [f"tile{i:04d}" for i in range(5)]
    assert res[0]["tiles_dr"] == (tiles[:3], True) and res[1]["tiles_dr"] == (tiles[3:], True)
    assert res[0]["tiles_dr_few"] == (["only"], False) and res[1]["tiles_dr_few"] == (["only"], False)
    assert res[0]["bits_tile_parallel"] == ([bits], False) and res[1]["bits_tile_parallel"] == ([bits], False)
    for r in range(world):
        assert res[r]["cards"] == {"ranks_reporting": 2,
                                   "sclk_mhz": {"lowest_rank_mean": 2080.0, "highest_rank_mean": 2140.0},
                                   "power_w": {"lowest_rank_mean": 1350.0, "highest_rank_mean": 1358.0},
                                   "hbm_c": {"lowest_rank_mean": 66.0, "highest_rank_mean": 71.0}}
        assert res[r]["cards_one_missing"]["ranks_reporting"] == 1
        assert res[r]["cards_one_missing"]["sclk_mhz"] == {"lowest_rank_mean": 2140.0, "highest_rank_mean": 2140.0}
        assert res[r]["cards_none"] is None
    pooled = np.concatenate([np.concatenate(res[r]["local"]) for r in range(world)]).astype(np.float32)
    for r in range(world):
        assert res[r]["count"] == pooled.size
        np.testing.assert_allclose(res[r]["p999"], np.percentile(pooled.astype(np.float64), 99.9), rtol=1e-12)
        np.testing.assert_allclose(res[r]["p50"], np.percentile(pooled.astype(np.float64), 50.0), rtol=1e-12)
        assert res[r]["median"] == float(np.median(pooled))
        assert res[r]["solo_median"] == res[r]["solo_ref"]
        assert res[r]["agg"] == (2.0, 30.0)
