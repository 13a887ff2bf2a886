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


# ---- build hygiene (round 4): the checks that would have caught round 3's plan-creation aborts on the CPU ------------

def test_host_stubs_and_code_object_name_the_same_kernels():
    """Every kernel name the host half registers with the HIP runtime exists in the gfx950 code object.  A library
    whose two halves were compiled from different states of a source aborts the process inside the runtime
    ("Cannot find Symbol with name ..."); this needs no GPU to see."""
    bc, hip = pkg("_buildcheck"), pkg("_hip")
    so = hip.library_path()
    host, dev = bc.host_registered_kernels(so), bc.device_kernels(so)
    assert len(dev) > 150 and len(host) > 150
    assert not (host - dev), sorted(host - dev)[:5]
    assert bc.kernel_mismatch(so) == []
    # the instantiations the bench runs at the headline geometry are among them
    for needle in ("k_x_inverse_split_chain", "k_y_forward_mirror_staticILi2187E", "k_x_inverse_dual_reg", "k_decode"):
        assert any(needle in k for k in dev), needle


def test_library_carries_the_digest_of_the_sources_in_the_tree():
    bc, hip = pkg("_buildcheck"), pkg("_hip")
    tree = bc.source_hash()
    assert bc.embedded_source_hash(hip.library_path()) == tree
    assert hip.lib().m3d_source_hash().decode() == tree


def test_stale_library_is_refused(monkeypatch):
    """A library built from other sources than the tree's is not loaded (no silent mismatch, no fallback)."""
    import pytest

    bc, hip = pkg("_buildcheck"), pkg("_hip")
    hip.lib()
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(bc, "source_hash", lambda *a, **k: "0" * 64)
    with pytest.raises(ImportError, match="was built from sources"):
        hip.lib()


def test_mismatched_halves_are_detected(tmp_path):
    """The detector itself: rename one registered kernel string in a copy of the library (what a source edited between
    hipcc's device and host passes produces) and the mismatch is reported by name."""
    bc, hip = pkg("_buildcheck"), pkg("_hip")
    blob = hip.library_path().read_bytes()
    victim = sorted(k for k in bc.host_registered_kernels(hip.library_path()) if "k_y_forward_mirror_static" in k)[0]
    forged = victim.replace("k_y_forward_mirror_static", "k_y_forward_mirror_statiX")
    fat = bc._Elf(blob).section(".hip_fatbin")   # the copies inside the code object stay: patch the host half only
    pos, patched = 0, bytearray(blob)
    while True:
        pos = blob.find(victim.encode() + b"\0", pos)
        if pos < 0:
            break
        if not (fat["off"] <= pos < fat["off"] + fat["size"]):
            patched[pos:pos + len(victim)] = forged.encode()
        pos += 1
    bad = tmp_path / "libm3d_forged.so"
    bad.write_bytes(bytes(patched))
    assert bc.kernel_mismatch(bad) == [forged]
