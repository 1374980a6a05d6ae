"""What the compiler made of each kernel: registers, spills, scratch and static LDS per kernel, read from the gfx950 code objects inside a
built libartgpu.so (the `.hip_fatbin` section holds one clang offload bundle per translation unit; each gfx950 entry is an ELF whose
NT_AMDGPU_METADATA note is a msgpack map with one record per kernel).  Host-side tooling for tests/test_kernel_resources.py and
scripts/kernel_resources.py -- a persistent kernel that owns a CU with sixteen waves has 128 registers per lane and nothing to hide a
scratch access behind, so a spill that creeps in with a source or toolchain change should fail a test here, not show up as a slower bench."""
from __future__ import annotations

import struct

_BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
_SHT_NOTE = 7
_NT_AMDGPU_METADATA = 32


def _sections(elf: bytes):
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + k * shentsize) for k in range(shnum)]
    strtab = secs[shstrndx][4]
    for s in secs:
        end = elf.index(b"\0", strtab + s[0])
        yield elf[strtab + s[0]:end].decode(), s[1], s[4], s[5]        # name, type, offset, size


def code_objects(library_path: str, arch: str = "gfx950"):
    """the device ELFs for `arch` inside a host shared library built by hipcc"""
    host = open(library_path, "rb").read()
    fat = [(off, size) for name, _, off, size in _sections(host) if name == ".hip_fatbin"]
    if not fat:
        raise ValueError(f"{library_path}: no .hip_fatbin section")
    blob = host[fat[0][0]:fat[0][0] + fat[0][1]]
    j = blob.find(_BUNDLE_MAGIC)
    while j >= 0:
        n, = struct.unpack_from("<Q", blob, j + len(_BUNDLE_MAGIC))
        o = j + len(_BUNDLE_MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, o)
            o += 24
            triple = blob[o:o + tlen].decode()
            o += tlen
            if size and triple.endswith(arch):
                yield blob[j + off:j + off + size]
        j = blob.find(_BUNDLE_MAGIC, j + 1)


def kernel_records(elf: bytes):
    """the `amdhsa.kernels` records of one device ELF"""
    import msgpack
    for _, typ, off, size in _sections(elf):
        if typ != _SHT_NOTE:
            continue
        o, end = off, off + size
        while o < end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, o)
            o += 12
            name = elf[o:o + namesz]
            o += (namesz + 3) & ~3
            desc = elf[o:o + descsz]
            o += (descsz + 3) & ~3
            if ntype == _NT_AMDGPU_METADATA and name.startswith(b"AMDGPU"):
                yield from msgpack.unpackb(desc, raw=False).get("amdhsa.kernels", [])


def demangle(names):
    import subprocess
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    return out[:len(names)]


def kernel_table(library_path: str):
    """{demangled kernel name: {vgprs, vgpr_spills, sgprs, sgpr_spills, scratch_bytes, static_lds_bytes, max_workgroup}}"""
    recs = [r for co in code_objects(library_path) for r in kernel_records(co)]
    table = {}
    for r, name in zip(recs, demangle([r[".name"] for r in recs])):
        table[name] = dict(vgprs=r[".vgpr_count"], vgpr_spills=r[".vgpr_spill_count"], sgprs=r[".sgpr_count"], sgpr_spills=r[".sgpr_spill_count"],
                           scratch_bytes=r[".private_segment_fixed_size"], static_lds_bytes=r[".group_segment_fixed_size"],
                           max_workgroup=r[".max_flat_workgroup_size"])
    return table
