"""CPU: bench.py's host logic - wall-budget step count, tile shares of the strong-scaling workload, and the
plain ``--gpus N`` start (N ranks as a child process, loud failure when the node has fewer GPUs)."""

import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def bench():
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        import bench as mod
    finally:
        sys.argv = argv
    return mod


def test_steps_are_an_upper_bound_under_the_wall_budget(bench):
    # the driver's command: 20 steps of ~37 s with ~400 s left -> what fits, never zero
    assert bench.steps_that_fit(20, 36.5, 400.0) == 10
    assert bench.steps_that_fit(20, 36.5, 10.0) == 1
    assert bench.steps_that_fit(3, 1.0, 400.0) == 3
    assert bench.steps_that_fit(4, 0.0, 5.0) == 4


def test_contiguous_tile_shares_cover_the_volume(bench):
    for tiles, world in ((64, 8), (8, 8), (10, 4), (7, 2), (8, 1)):
        shares = [list(bench.contiguous_share(tiles, world, r)) for r in range(world)]
        assert sum(shares, []) == list(range(tiles))
        assert max(map(len, shares)) - min(map(len, shares)) <= 1


def test_plain_multi_gpu_start_fails_loudly_without_the_gpus():
    """WORLD_SIZE unset and --gpus 2 on a node with fewer GPUs: no silent one-GPU measurement."""
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("node has the GPUs")
    env = {k: v for k, v in __import__("os").environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "only" in p.stderr and "GPU" in p.stderr
    assert not any(line.startswith("{") for line in p.stdout.splitlines())


def test_world_size_must_match_gpus():
    env = dict(__import__("os").environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr


def test_product_path_never_touches_the_oracle():
    """oracle/ is test infrastructure: no file of the package names it, and bench.py imports it in cpu_baseline only."""
    import ast

    pkg_dir = ROOT / "merfish3d-analysis_amd"
    for path in list(pkg_dir.rglob("*.py")) + [ROOT / "merfish3d_analysis_amd.py"]:
        assert "oracle" not in path.read_text().lower(), path
    tree = ast.parse((ROOT / "bench.py").read_text())
    importers = set()
    for fn in [n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.Module))]:
        for node in (fn.body if isinstance(fn, ast.Module) else ast.walk(fn)):
            if isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle":
                importers.add(getattr(fn, "name", "<module>"))
            if isinstance(node, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in node.names):
                importers.add(getattr(fn, "name", "<module>"))
    assert importers <= {"cpu_baseline"}, importers


def test_visible_gpu_count_takes_the_smallest_claim_and_admits_ignorance(monkeypatch):
    """several visibility variables -> the smallest list counts; no list and no readable KFD
    topology -> None (the launcher then skips its pre-check instead of refusing a usable machine)."""
    import sys

    sys.argv = ["bench.py"]
    import bench

    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setattr(bench.Path, "glob", lambda self, pat: iter(()))      # no topology to read
    assert bench.visible_gpu_count() is None
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2,3")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0,1")
    assert bench.visible_gpu_count() == 2
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_count() == 0


def test_card_state_sampler_reads_the_hwmon_files_and_survives_their_absence(bench, tmp_path):
    """The diagnostic next to ``value`` (clocks, board power, temperatures while the timed region runs): amdgpu's hwmon
    units, a running summary, and no record - not an error - where the files are missing."""
    import time

    hw = tmp_path / "hwmon3"
    hw.mkdir()
    for name, raw in {"freq1_input": 2140000000, "freq2_input": 2000000000, "power1_input": 1350000000,
                      "power1_cap": 1400000000, "temp2_input": 56000, "temp3_input": 70000}.items():
        (hw / name).write_text(f"{raw}\n")
    card = bench.CardState(hw, period_s=0.02).start()
    time.sleep(0.2)
    (hw / "freq1_input").write_text("1900000000\n")
    time.sleep(0.2)
    running = card.summary()          # usable while the sampler runs (one JSON line per timed step)
    card.stop()
    got = card.summary()
    assert running["samples"] >= 2 and got["samples"] >= running["samples"]
    assert got["sclk_mhz"]["max"] == 2140.0 and got["sclk_mhz"]["min"] == 1900.0
    assert got["mclk_mhz"]["mean"] == 2000.0 and got["power_w"]["mean"] == 1350.0 and got["power_cap_w"] == 1400.0
    assert got["junction_c"]["max"] == 56.0 and got["hbm_c"]["max"] == 70.0

    (hw / "temp3_input").unlink()     # a file that a kernel version does not have: the others are still reported
    partial = bench.CardState(hw, period_s=0.02).start()
    time.sleep(0.1)
    partial.stop()
    assert "hbm_c" not in partial.summary() and "sclk_mhz" in partial.summary()
    nothing = bench.CardState(None).start()
    nothing.stop()
    assert nothing.summary() is None
    assert bench.CardState.hwmon_of(0, sysfs=str(tmp_path / "no_such_sysfs")) is None
