"""CPU: libm3d_hip.so loads and exports every function include/*.h declares."""

import re

from conftest import ROOT, pkg


def _declared_functions():
    names = set()
    for header in sorted((ROOT / "include").glob("*.h")):
        text = re.sub(r"/\*.*?\*/", "", header.read_text(), flags=re.S)
        names |= set(re.findall(r"\b(m3d_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    hip = pkg("_hip")
    handle = hip.lib()
    declared = _declared_functions()
    assert len(declared) >= 10
    missing = [name for name in declared if not hasattr(handle, name)]
    assert not missing, missing
    assert set(hip.EXPORTED_SYMBOLS) == set(declared)
    assert handle.m3d_version() >= 100
    assert handle.m3d_status_string(-1).decode() == "invalid argument"


def test_geometry_entry_point_matches_reference_rule():
    hip = pkg("_hip")
    rl = pkg("utils.rlgc")
    handle = hip.lib()
    assert handle.m3d_next_fft_size(2098) == 2187
    padded, before, after = hip.i3((0, 0, 0)), hip.i3((0, 0, 0)), hip.i3((0, 0, 0))
    rc = handle.m3d_rlgc_geometry(hip.i3((64, 2048, 2048)), hip.i3((64, 51, 51)), 1, padded, before, after)
    assert rc == 0 and list(padded) == [128, 2187, 2187]
    pads = rl.pad_widths_zyx((64, 2048, 2048), (64, 51, 51))
    assert [tuple(p) for p in pads] == list(zip(before, after))
    assert handle.m3d_rlgc_geometry(hip.i3((0, 4, 4)), hip.i3((1, 1, 1)), 1, padded, before, after) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    hip = pkg("_hip")
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "_LIB_PATH", tmp_path / "nope.so")
    import pytest

    with pytest.raises(ImportError, match="no CPU fallback"):
        hip.lib()
