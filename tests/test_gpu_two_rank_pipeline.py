"""GPU, two ranks: DataRegistration -> PixelDecoder over a ``gloo`` group of two processes that share the box's one
GPU and one on-disk qi2lab datastore (write-behind on), against the same pipeline in one process on a copy of the
dataset: identical normalisation vectors and an identical filtered transcript table.  What the N > 1 job adds is then
exercised on device data: tiles dealt to ranks in both operators, the pooled radix-select all-reduces of the global
normalisation, the all-reduced mis-registration count, per-tile tables and images meeting in the shared directory,
rank 0's filter.  (RCCL itself carries these collectives at world size 1 in tests/test_gpu_rccl_single_rank.py; more
than one GPU is the driver's run.)  Reference: DataRegistration.py:2137-2156, PixelDecoder.py:981-1199, 4759-4870."""

import json
import os
import shutil
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def _result(stdout: str) -> dict:
    lines = [ln for ln in stdout.splitlines() if ln.startswith("RESULT ")]
    assert lines, stdout[-3000:]
    return json.loads(lines[-1][len("RESULT "):])


def test_two_ranks_on_one_datastore_equal_one_process(tmp_path):
    from test_gpu_pipeline import _experiment

    store, truth, _shifts, predictors = _experiment(zarr_root=tmp_path / "one")
    np.savez(tmp_path / "one" / "test_predictors.npz", **{f"{t}__{b}": v for (t, b), v in predictors.items()})
    del store
    shutil.copytree(tmp_path / "one", tmp_path / "two")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child = str(ROOT / "tests" / "dist_pipeline_child.py")
    single = subprocess.run([sys.executable, child, str(tmp_path / "one")], env={k: v for k, v in env.items()
                                                                               if k not in ("WORLD_SIZE", "RANK")},
                            capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert single.returncode == 0, single.stdout[-2000:] + single.stderr[-4000:]
    one = _result(single.stdout)
    pair = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), child, str(tmp_path / "two")],
                          env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert pair.returncode == 0, pair.stdout[-2000:] + pair.stderr[-4000:]
    two = _result(pair.stdout)
    print(f"\none process {one}\ntwo ranks   {two}")
    assert one["transcripts"] > 50 and min(one["per_tile"]) > 10
    assert two["suspect_all_ranks"] == one["suspect_all_ranks"] == 0
    assert two["resident_left"] == 0 and two["pending_left"] == 0       # evicted per tile, flushed before returning
    assert two["normalization"] == one["normalization"] and two["background"] == one["background"]
    assert two["per_tile"] == one["per_tile"] and two["transcripts"] == one["transcripts"]
    assert two["table_digest"] == one["table_digest"]                   # the same rows
    assert len(truth) > 0
