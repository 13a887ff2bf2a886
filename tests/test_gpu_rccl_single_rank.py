"""The multi-rank code path on the hardware a one-GPU box has: fresh child processes that join an ``nccl``
(= RCCL) process group of one rank on ``cuda:0`` before any other GPU call.

* ``PixelDecoder._global_normalization_vectors`` with ``kernels._dist_ready()`` true: ``m3d_radix_hist`` on the
  device, ``dist.all_reduce`` of the device int64[256] histogram per radix pass, bit-equal to the local
  (``group=False``) route and equal to the reference run (PixelDecoder.py:981-1199, 1013-1020).
* ``bench.aggregate_over_ranks`` / ``agree_min`` on device tensors.
* ``bench.py --workload config3_small --tiles 2 --gpus 1`` with the RCCL group forced on: the strong-scaling
  workload (per-round registration, tiles dealt to ranks: PixelDecoder.py:4811-4838) end to end.
"""

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.gpu


def _env(**extra):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra)
    return env


def test_global_normalization_runs_its_all_reduce_on_rccl():
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "rccl_child.py")], env=_env(), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RCCL_CHILD ")][-1]
    out = json.loads(line[len("RCCL_CHILD "):])
    assert out["backend"] == "nccl" and out["world"] == 1 and out["dist_ready"]
    # 16 bits x (count + 4 passes per order statistic) of the two pooled medians: hundreds of collectives
    assert out["collective_calls"] >= 16 * 2 * 5 and out["collective_calls_local_route"] == 0
    assert out["collective_on_device"] and out["collective_shapes"] == ["((256,), 'torch.int64')"]
    assert out["norm_bits_equal"] and out["bkd_bits_equal"]
    assert out["norm_vs_reference_run"] < 2e-5 and out["bkd_vs_reference_run"] < 2e-5
    assert out["pool_count_ok"] and out["pool_median_ok"]
    assert out["aggregate"] == [1.25, 37.0] and out["agree_min"] == 11


def test_bench_config3_small_through_the_rccl_group():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "config3_small", "--tiles", "2",
                        "--gpus", "1", "--no-cpu-baseline"],
                       env=_env(M3D_BENCH_FORCE_DIST="1", M3D_DIST_SINGLE_RANK="1", M3D_BENCH_WALL_S="600"),
                       capture_output=True, text=True, timeout=1200, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    rec = json.loads(lines[-1])
    assert rec["process_group"] == {"backend": "nccl", "world_size": 1}
    assert rec["scaling"] == "strong" and rec["n_gpus"] == 1
    assert rec["config"]["tiles_timed"] == 2 and rec["steps"] == 2
    assert rec["registration"]["calls_rank0"] == 2 * 7          # 8 rounds: 7 moving rounds per tile
    assert rec["registration"]["max_shift_error_px"] < 0.35     # the reference's own tolerance (+-0.35 px)
    assert rec["transcripts_rank0"] > 0 and rec["value"] > 0
    assert rec["roofline"]["frac"] > 0


def test_bench_config3_small_out_of_and_back_into_a_datastore_directory():
    """``bench.py --store disk`` (round 4) at rehearsal size, through the one-rank RCCL group: every tile's images come
    from and go back to an on-disk qi2lab datastore with read-ahead and write-behind under the compute; the record says
    what the main thread waited for, what the I/O threads did, and that what is on disk is what was computed."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "config3_small", "--tiles", "3",
                        "--gpus", "1", "--no-cpu-baseline", "--store", "disk", "--io-read-threads", "2",
                        "--io-write-threads", "4"],
                       env=_env(M3D_BENCH_FORCE_DIST="1", M3D_DIST_SINGLE_RANK="1", M3D_BENCH_WALL_S="600"),
                       capture_output=True, text=True, timeout=1200, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["config"]["tiles_timed"] == 3 and rec["steps"] == 3 and rec["transcripts_rank0"] > 0
    st = rec["store"]
    assert st["round_trip_equal"] is True
    assert st["io_threads"] == {"read_ahead_zstd": 2, "write_behind_zstd": 4, "python_io_threads": 2}
    assert st["value_including_io"] == rec["value"] and st["value_compute_only"] >= st["value_including_io"]
    assert st["per_tile_counts"]["images_read"] >= 8 + 16 + 16          # 8 fiducial + 16 readout + 16 predictor images
    assert st["per_tile_counts"]["images_written"] >= 8 + 16            # deconvolved fiducials and readouts
    assert set(st["caller_wait_s_per_tile"]) <= {"read_wait", "write_wait"}
    assert rec["registration"]["all_ranks"] == {"calls": 3 * 7, "off_by_more_than_px": 0.35, "rounds_off": 0}
