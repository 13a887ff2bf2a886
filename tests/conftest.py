"""pytest configuration: `gpu` marker, repo-root imports, shared loaders."""

from __future__ import annotations

import importlib
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch

        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pkg(name: str = ""):
    """Import ``merfish3d-analysis_amd[.name]`` (the directory name has a hyphen)."""
    return importlib.import_module("merfish3d-analysis_amd" + (f".{name}" if name else ""))


@pytest.fixture(scope="session")
def golden():
    def _load(name: str):
        return np.load(GOLDEN / f"{name}.npz", allow_pickle=False)

    return _load


@pytest.fixture(scope="session", autouse=True)
def _oracle_built():
    from oracle import build as oracle_build

    oracle_build.build()
