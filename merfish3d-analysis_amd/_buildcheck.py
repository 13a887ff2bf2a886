"""Self-consistency checks of ``csrc/libm3d_hip.so`` that need no GPU.

Two questions, both answered from the file alone:

* *Was the library built from the sources that are in the tree?*  ``source_hash`` is a SHA-256 over the
  bytes of every ``csrc/*.hip``, ``csrc/*.h`` and ``include/*.h``; ``csrc/build.py`` compiles from a
  snapshot of exactly those bytes and links the digest in (``m3d_source_hash()``).  ``_hip.lib()``
  compares the two at load time and refuses a stale library.
* *Do the host half and the device half of the library agree?*  hipcc reads a ``.hip`` file once per
  pass (device, then host).  A source edited between the passes yields host stubs that register
  kernel names the gfx950 code object does not hold; the HIP runtime then aborts the process on the
  first ``hipFuncSetAttribute`` / launch of such a kernel (round 3).
  ``host_registered_kernels`` / ``device_kernels`` read both halves so that a test and the build can
  compare them before a GPU minute is spent.

Pure Python (struct + a minimal ELF64 reader): no binutils / LLVM tools needed on the box.
"""

from __future__ import annotations

import hashlib
import re
import struct
from pathlib import Path

PACKAGE_DIR = Path(__file__).resolve().parent
CSRC = PACKAGE_DIR / "csrc"
INCLUDE = PACKAGE_DIR.parent / "include"

_BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
_MANGLED = re.compile(rb"_Z[A-Za-z0-9_.$]+\Z")


def source_files(csrc: Path = CSRC, include: Path = INCLUDE) -> list[tuple[str, Path]]:
    """(name inside the snapshot, path) of every file the library is compiled from, sorted."""
    files = [(f"csrc/{p.name}", p) for p in csrc.glob("*.hip")]
    files += [(f"csrc/{p.name}", p) for p in csrc.glob("*.h")]
    files += [(f"include/{p.name}", p) for p in include.glob("*.h")]
    return sorted(files)


def hash_sources(blobs: list[tuple[str, bytes]]) -> str:
    h = hashlib.sha256()
    for name, data in sorted(blobs):
        h.update(name.encode() + b"\0" + struct.pack("<Q", len(data)) + data)
    return h.hexdigest()


def source_hash(csrc: Path = CSRC, include: Path = INCLUDE) -> str:
    return hash_sources([(name, path.read_bytes()) for name, path in source_files(csrc, include)])


# ---------------------------------------------------------------------------------------------------
# minimal ELF64 little-endian reader


class _Elf:
    def __init__(self, data: bytes | memoryview):
        self.d = memoryview(data)
        if bytes(self.d[:4]) != b"\x7fELF" or self.d[4] != 2 or self.d[5] != 1:
            raise ValueError("not a little-endian ELF64 image")
        (self.shoff,) = struct.unpack_from("<Q", self.d, 0x28)
        self.shentsize, self.shnum, self.shstrndx = struct.unpack_from("<HHH", self.d, 0x3A)
        self.sections = []
        for i in range(self.shnum):
            name, typ, _flags, _addr, off, size, link, _info, _align, entsize = struct.unpack_from(
                "<IIQQQQIIQQ", self.d, self.shoff + i * self.shentsize)
            self.sections.append(dict(name_off=name, type=typ, off=off, size=size, link=link, entsize=entsize))
        strtab = self.sections[self.shstrndx]
        for s in self.sections:
            s["name"] = self._cstr(strtab["off"] + s["name_off"])

    def _cstr(self, off: int) -> str:
        end = off
        while self.d[end] != 0:
            end += 1
        return bytes(self.d[off:end]).decode(errors="replace")

    def section(self, name: str):
        for s in self.sections:
            if s["name"] == name:
                return s
        return None

    def section_bytes(self, name: str) -> bytes:
        s = self.section(name)
        if s is None or s["type"] == 8:  # SHT_NOBITS
            return b""
        return bytes(self.d[s["off"]:s["off"] + s["size"]])

    def symbols(self) -> set[str]:
        names = set()
        for s in self.sections:
            if s["type"] not in (2, 11):  # SHT_SYMTAB, SHT_DYNSYM
                continue
            strs = self.sections[s["link"]]
            for i in range(s["size"] // 24):
                (st_name,) = struct.unpack_from("<I", self.d, s["off"] + i * 24)
                if st_name:
                    names.add(self._cstr(strs["off"] + st_name))
        return names


def host_registered_kernels(so_path: Path) -> set[str]:
    """Mangled kernel names the host half passes to ``__hipRegisterFunction``: the NUL-terminated
    ``_Z...`` strings of the read-only data sections."""
    elf = _Elf(Path(so_path).read_bytes())
    names: set[str] = set()
    for s in elf.sections:
        if not s["name"].startswith(".rodata"):
            continue
        for piece in elf.section_bytes(s["name"]).split(b"\0"):
            if _MANGLED.match(piece):
                names.add(piece.decode())
    return names


def device_code_objects(so_path: Path, arch: str = "gfx950") -> list[bytes]:
    """The ``arch`` code objects of every offload bundle in ``.hip_fatbin`` (one bundle per
    translation unit, concatenated by the linker)."""
    elf = _Elf(Path(so_path).read_bytes())
    fat = elf.section_bytes(".hip_fatbin")
    if not fat:
        raise ValueError(f"{so_path}: no .hip_fatbin section")
    out = []
    pos = fat.find(_BUNDLE_MAGIC)
    if pos < 0:
        raise ValueError(f"{so_path}: .hip_fatbin holds no uncompressed offload bundle "
                         f"(starts with {fat[:4]!r}; build without --offload-compress)")
    while pos >= 0:
        (count,) = struct.unpack_from("<Q", fat, pos + len(_BUNDLE_MAGIC))
        cur = pos + len(_BUNDLE_MAGIC) + 8
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", fat, cur)
            triple = fat[cur + 24:cur + 24 + tlen].decode()
            cur += 24 + tlen
            if "amdgcn" in triple and triple.rstrip("-").endswith(arch) and size:
                out.append(fat[pos + off:pos + off + size])
        pos = fat.find(_BUNDLE_MAGIC, pos + 1)
    return out


def device_kernels(so_path: Path, arch: str = "gfx950") -> set[str]:
    """Names of the kernels in the library's code objects (symbols that own a ``.kd`` descriptor)."""
    names: set[str] = set()
    for blob in device_code_objects(so_path, arch):
        for sym in _Elf(blob).symbols():
            if sym.endswith(".kd"):
                names.add(sym[:-3])
    return names


def kernel_mismatch(so_path: Path, arch: str = "gfx950") -> list[str]:
    """Kernel names the host half registers that the device half does not define (must be empty)."""
    return sorted(host_registered_kernels(so_path) - device_kernels(so_path, arch))


def embedded_source_hash(so_path: Path) -> str | None:
    """The digest ``csrc/build.py`` linked in, read from the file (no dlopen)."""
    m = re.search(rb"m3d-source-sha256:([0-9a-f]{64})", Path(so_path).read_bytes())
    return m.group(1).decode() if m else None
