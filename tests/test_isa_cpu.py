"""Guards on the BUILT device code (CPU only: the gfx950 code objects are pulled out of libgreb_hip.so and read with
llvm-objdump / llvm-readelf).  greb_chain6.h hand-schedules the chain sweep inside one asm statement, where the
compiler's hazard recogniser does not look: a VGPR written by a VALU instruction must not be read through DPP by either
of the next two instructions.  Parity tests would catch a broken sweep only on the GPU; this catches a toolchain or
edit that closes the gap before anything runs.  Also: the FAST kernels that sit at a register cliff stay off scratch."""
import os
import re
import shutil
import struct
import subprocess
import tempfile

import pytest

from greb_climate_model_amd import build

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _tool(name):
    p = os.path.join(LLVM, name)
    return p if os.path.exists(p) else shutil.which(name)


@pytest.fixture(scope="module")
def code_objects():
    """Every gfx950 code object of the release library (one offload bundle per source file), as temp files."""
    objcopy, objdump = _tool("llvm-objcopy"), _tool("llvm-objdump")
    if not (objcopy and objdump and _tool("llvm-readelf")):
        pytest.skip("llvm binutils not present")
    lib = build.build_lib()
    d = tempfile.mkdtemp(prefix="greb_isa_")
    fat = os.path.join(d, "fat.bin")
    subprocess.run([objcopy, "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    blob = open(fat, "rb").read()
    out, pos = [], blob.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                path = os.path.join(d, f"co{len(out)}.co")
                open(path, "wb").write(blob[pos + off:pos + off + size])
                out.append(path)
        pos = blob.find(MAGIC, pos + 1)
    assert len(out) >= len([s for s in build.SOURCES if s.endswith(".hip")]), out
    yield out
    shutil.rmtree(d, ignore_errors=True)


def _functions(path):
    """{demangled name: [instruction text, ...]} of one code object."""
    txt = subprocess.run([_tool("llvm-objdump"), "-d", "--demangle", path], check=True, capture_output=True, text=True).stdout
    fns, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:\s*$", line)
        if m:
            cur = fns.setdefault(m.group(1), [])
            continue
        if cur is not None and line.startswith("\t"):
            ins = line.split("//")[0].strip()
            if ins:
                cur.append(ins)
    return fns


def _vregs(op):
    m = re.fullmatch(r"v(\d+)", op)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def _written(ins):
    """VGPRs a VALU / DS / VMEM instruction writes (its first operand), or the empty set."""
    m = re.match(r"^(v_\S+|ds_read\S*|global_load\S*|buffer_load\S*)\s+([^,\s]+)", ins)
    if not m or m.group(1).startswith(("v_cmp", "v_cmpx")):
        return set()
    return _vregs(m.group(2))


def test_dpp_reads_are_two_instructions_behind_the_write(code_objects):
    """Every DPP instruction of the library: its DPP source (src0) was not written by the two instructions before it
    (an s_nop k counts as k + 1).  The 36-instruction sweep of greb_chain6.h keeps three."""
    n_dpp = n_sweep = 0
    for path in code_objects:
        for name, body in _functions(path).items():
            slots = []  # one entry per wait state: the registers written there
            for ins in body:
                if re.search(r"\b(row_ror|row_shr|row_shl|wave_ror|wave_rol|wave_shr|wave_shl|quad_perm|row_mirror|row_half_mirror|row_bcast)", ins):
                    ops = [o.strip() for o in re.sub(r"^\S+\s+", "", ins).split(",")]
                    src0 = _vregs(ops[1].split()[0]) if len(ops) > 1 else set()
                    recent = set().union(*slots[-2:]) if slots else set()
                    assert not (src0 & recent), (name[:80], ins, "DPP source written within the last two instructions")
                    n_dpp += 1
                    n_sweep += "v_fmac_f32_dpp" in ins or "v_subrev_f32_dpp" in ins
                m = re.match(r"^s_nop\s+(\d+)", ins)
                if m:
                    slots.extend([set()] * (int(m.group(1)) + 1))
                else:
                    slots.append(_written(ins))
    assert n_dpp > 500 and n_sweep > 50, (n_dpp, n_sweep)  # the sweeps are in there and were looked at


def _kernel_notes(path):
    txt = subprocess.run([_tool("llvm-readelf"), "--notes", path], check=True, capture_output=True, text=True).stdout
    out = {}
    for blk in re.split(r"\n\s+- \.", txt):
        name = re.search(r"\.name:\s+(\S+)", blk)
        if name and ".vgpr_count" in blk:
            out[name.group(1)] = {k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
                                  for k in ("vgpr_count", "private_segment_fixed_size", "vgpr_spill_count")}
    return out


def test_fast_kernels_at_a_register_cliff_stay_off_scratch(code_objects):
    notes = {}
    for path in code_objects:
        notes.update(_kernel_notes(path))
    member = {k: v for k, v in notes.items() if "member_kernel" in k and "ILb0E" in k}  # member_kernel<false, ...>: FAST
    assert member, sorted(notes)[:5]
    for k, v in member.items():
        assert v["private_segment_fixed_size"] == 0 and v["vgpr_spill_count"] == 0 and v["vgpr_count"] <= 256, (k, v)
    rows = {k: v for k, v in notes.items() if "dif_rows_kernel" in k and "ILb0E" in k}
    assert rows
    for k, v in rows.items():  # four waves per SIMD need <= 128 registers (greb_climate_model_amd/build.py: EXTRA_FLAGS)
        assert v["private_segment_fixed_size"] == 0 and v["vgpr_count"] <= 128, (k, v)
    for path in code_objects:
        for name, body in _functions(path).items():
            if ("member_kernel<false" in name or "dif_rows_kernel<false" in name) and not name.startswith("__"):
                assert not [i for i in body if i.startswith("scratch_")], name[:80]


def test_row_strip_kernels_load_their_constants_through_the_scalar_cache(code_objects):
    """The row-strip kernels count vmcnt by hand (greb_rows.h): LDS-DMA and stores only.  A plain vector load in there
    -- the per-row constants of step_rows_kernel once were four global_load_dword + s_waitcnt vmcnt(0) per row, because
    the table was read through a generic pointer behind the kernel's own stores -- drains the whole LDS-DMA ring every
    row.  Tables and task words must come by s_load (constant address space) or by value."""
    seen = 0
    for path in code_objects:
        for name, body in _functions(path).items():
            if not re.search(r"(step_rows_kernel|dif_rows_kernel|circ_rows_kernel)<", name) or name.startswith("__"):
                continue
            seen += 1
            bad = [i for i in body if re.match(r"^(global_load_(dword|dwordx2|dwordx3|dwordx4|ubyte|ushort|sbyte|sshort)|flat_load\S*|buffer_load\S*)\s", i)]
            if "circ_rows_kernel<" in name:  # its flag polls: one dword per dependency, agent scope, waited for where they are issued
                bad = [i for i in bad if not re.match(r"^global_load_dword v\d+, v\[\d+:\d+\], off sc1$", i)]
            assert not bad, (name[:90], bad[:4])
            if "step_rows_kernel<false" in name or "circ_rows_kernel<false" in name:
                assert sum(i.startswith("s_load_dword") for i in body) >= 8, name[:90]  # arguments, task, row constants
    assert seen >= 7, seen


def test_pending_lds_reads_are_left_alone_until_their_wait(code_objects):
    """greb_rows.h splits a row read into read_pair_issue (six ds_read_b64) and read_pair_finish (s_waitcnt lgkmcnt(0))
    with the window shift between them, in two asm statements: nothing but register allocation keeps the compiler from
    copying a destination register in between, which would read data that has not arrived.  Checked on the built code:
    between a ds_read_* and the next lgkmcnt(0) wait no instruction reads or writes its destination VGPRs."""
    checked = 0
    for path in code_objects:
        for name, body in _functions(path).items():
            if "_rows_kernel<" not in name or name.startswith("__"):
                continue
            pending = set()
            for ins in body:
                if ins.startswith("s_waitcnt") and ("lgkmcnt(0)" in ins or ins.strip() == "s_waitcnt 0"):
                    pending = set()
                    continue
                if re.match(r"^s_waitcnt\s+vmcnt\(\d+\)\s+lgkmcnt\(0\)", ins) or "lgkmcnt(0)" in ins:
                    pending = set()
                    continue
                m = re.match(r"^(\S+)\s+(.*)$", ins)
                regs = set()
                if m:
                    for op in re.split(r"[,\s]+", m.group(2)):
                        regs |= _vregs(op.strip())
                if pending and not ins.startswith("ds_read"):
                    assert not (regs & pending), (name[:80], ins, sorted(regs & pending))
                if ins.startswith("ds_read") or ins.startswith("ds_bpermute") or ins.startswith("ds_permute"):
                    if pending:  # a second read may not overwrite or consume a pending one either
                        srcs = set()
                        for op in re.split(r"[,\s]+", m.group(2))[1:]:
                            srcs |= _vregs(op.strip())
                        assert not (srcs & pending), (name[:80], ins)
                    pending |= _written(ins) if ins.startswith("ds_read") else _vregs(re.split(r"[,\s]+", m.group(2))[0])
                    checked += 1
    assert checked > 100, checked


def test_every_device_header_is_a_build_dependency():
    """An edit to any csrc/*.h must trigger a rebuild (a stale libgreb_hip.so once shipped after an edit to
    greb_chain6.h alone)."""
    csrc = os.path.join(os.path.dirname(build.__file__), "csrc")
    headers = sorted(f for f in os.listdir(csrc) if f.endswith(".h"))
    listed = {os.path.basename(h) for h in build.HEADERS}
    assert headers and set(headers) <= listed, sorted(set(headers) - listed)
    sources = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".cpp")))
    assert set(sources) == set(build.SOURCES), (sources, build.SOURCES)
