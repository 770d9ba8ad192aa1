"""Which machine code did a measurement run on?  sha256 of the gfx950 code of one kernel as it sits in libgreb_hip.so.

The counter passes behind bench.py's `roofline.traffic` are collected by rocprofv3 in separate runs and committed under
profiles/; a kernel edit after that would silently keep the old ratio.  Every such record therefore carries the hash
of the profiled kernel's code (tools/pmc_summary.py, tools/pmc_rows_summary.py), bench.py compares it with the library it
has loaded (`traffic_source.matches_loaded_library`), and tests/test_profiles_cpu.py fails when they differ.

Pure Python (no binutils needed on the GPU box): the host ELF's .hip_fatbin section holds one clang offload bundle per
source file; each bundle's gfx950 entry is an ELF whose .symtab names the kernels (mangled) with address and size."""
from __future__ import annotations

import hashlib
import struct

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _sections(blob: bytes):
    """{name: (offset, size, addr, link)} and a list of raw headers of an ELF64 little-endian image."""
    if blob[:4] != b"\x7fELF" or blob[4] != 2 or blob[5] != 1:
        raise ValueError("not a little-endian ELF64 image")
    shoff, = struct.unpack_from("<Q", blob, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
    raw = []
    for i in range(shnum):
        name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from("<IIQQQQIIQQ", blob, shoff + i * shentsize)
        raw.append((name, typ, addr, off, size, link, entsize))
    stroff = raw[shstrndx][3]
    out = {}
    for name, typ, addr, off, size, link, entsize in raw:
        end = blob.index(b"\0", stroff + name)
        out[blob[stroff + name:end].decode()] = (off, size, addr, link, typ, entsize)
    return out, raw


def gfx950_code_objects(lib_path: str) -> list[bytes]:
    blob = open(lib_path, "rb").read()
    secs, _ = _sections(blob)
    if ".hip_fatbin" not in secs:
        raise ValueError(f"{lib_path}: no .hip_fatbin section")
    off, size = secs[".hip_fatbin"][:2]
    fat = blob[off:off + size]
    out, pos = [], fat.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", fat, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            o, s, tl = struct.unpack_from("<QQQ", fat, q)
            triple = fat[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and s:
                out.append(fat[pos + o:pos + o + s])
        pos = fat.find(MAGIC, pos + 1)
    return out


def kernel_functions(lib_path: str, pattern: str) -> dict[str, bytes]:
    """{mangled name: machine code} of every FUNC symbol of the library's gfx950 code whose name contains `pattern`
    (a piece of the MANGLED name, e.g. 'diffusion_stream_kernelILb0ELi96ELi48E')."""
    found = {}
    for co in gfx950_code_objects(lib_path):
        secs, raw = _sections(co)
        if ".symtab" not in secs:
            continue
        off, size, _, link, _, entsize = secs[".symtab"]
        stroff = raw[link][3]
        for i in range(size // (entsize or 24)):
            name, info, other, shndx, value, sz = struct.unpack_from("<IBBHQQ", co, off + i * (entsize or 24))
            if (info & 0xF) != 2 or sz == 0 or shndx == 0 or shndx >= len(raw):  # STT_FUNC, defined
                continue
            end = co.index(b"\0", stroff + name)
            nm = co[stroff + name:end].decode()
            if pattern not in nm:
                continue
            _, _, saddr, soff, _, _, _ = raw[shndx]
            start = soff + (value - saddr)
            found[nm] = co[start:start + sz]
    return found


def code_sha(lib_path: str, pattern: str) -> str | None:
    """sha256 over (name, code) of the matching kernels in name order; None when nothing matches."""
    fns = kernel_functions(lib_path, pattern)
    if not fns:
        return None
    h = hashlib.sha256()
    for nm in sorted(fns):
        h.update(nm.encode()); h.update(b"\0"); h.update(fns[nm])
    return h.hexdigest()


def record(lib_path: str, pattern: str) -> dict:
    """The `code` entry of a profiles/*.json record."""
    return {"kernel_symbol_contains": pattern, "sha256": code_sha(lib_path, pattern),
            "of": "the kernel's gfx950 machine code in libgreb_hip.so (greb_climate_model_amd/codesha.py)"}


def source_check(rec: dict | None, file: str, lib_path: str) -> dict:
    """bench.py's `traffic_source`: the committed record, the hash it carries and whether the loaded library has that code."""
    out = {"file": file, "code_sha": None, "matches_loaded_library": None}
    if not rec or "code" not in rec:
        return out
    out["code_sha"] = rec["code"].get("sha256")
    try:
        out["matches_loaded_library"] = code_sha(lib_path, rec["code"]["kernel_symbol_contains"]) == out["code_sha"]
    except (OSError, ValueError, KeyError, struct.error):
        out["matches_loaded_library"] = None
    return out


if __name__ == "__main__":
    import sys
    from . import build
    for pat in sys.argv[1:]:
        print(pat, code_sha(build.LIB, pat), sorted(kernel_functions(build.LIB, pat)))
