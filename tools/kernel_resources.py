"""Resource usage of every gfx950 kernel in a built libtoppra_hip.so, read from the code objects' own metadata (no compile,
no GPU): the .hip_fatbin section holds one clang offload bundle per translation unit, each bundle an amdgcn ELF whose
NT_AMDGPU_METADATA note (msgpack) lists per kernel the vector / accumulator registers, the scratch ("private segment") bytes
per lane, the LDS bytes per block and the spilled scalar registers.

    python tools/kernel_resources.py [lib.so] [name-substring]

tests/test_kernel_resources.py asserts the numbers of the certified lane kernels against committed ceilings: a toolchain
that brings back the divergent regions of DESIGN.md section 3.2 (a conditionally-needed load sunk into a branch: 1.4 - 2.4 KB
of scratch per lane, 5 x slower, and the trigger of the wrong results of section 9) fails a CPU test instead of
silently costing 5 x or returning wrong bits."""
import os
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _section(path, name):
    """(offset, size) of an ELF64 section by name."""
    with open(path, "rb") as f:
        hdr = f.read(64)
        shoff, = struct.unpack_from("<Q", hdr, 0x28)
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", hdr, 0x3A)
        f.seek(shoff)
        table = f.read(shentsize * shnum)
        secs = [struct.unpack_from("<IIQQQQIIQQ", table, i * shentsize) for i in range(shnum)]
        f.seek(secs[shstrndx][4])
        strtab = f.read(secs[shstrndx][5])
        for s in secs:
            nm = strtab[s[0]:strtab.index(b"\0", s[0])]
            if nm == name.encode():
                return s[4], s[5]
    raise KeyError(name)


def code_objects(lib):
    """The amdgcn ELF images bundled into `lib`."""
    off, size = _section(lib, ".hip_fatbin")
    with open(lib, "rb") as f:
        f.seek(off)
        blob = f.read(size)
    out, pos = [], 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            eoff, esize, tsize = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tsize].decode()
            p += 24 + tsize
            if "amdgcn" in triple and esize:
                out.append(blob[pos + eoff:pos + eoff + esize])
        pos += len(MAGIC)
    return out


def _notes(elf):
    """NT_AMDGPU_METADATA (type 32) payloads of an ELF64 image."""
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
    for i in range(shnum):
        sh = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize)
        if sh[1] != 7:  # SHT_NOTE
            continue
        p, end = sh[4], sh[4] + sh[5]
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz]
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if ntype == 32 and name.startswith(b"AMDGPU"):
                yield desc


def kernels(lib=None):
    """{demangled-ish kernel name: dict(vgpr, agpr, scratch, lds, sgpr_spill, vgpr_spill)} of every kernel in the library."""
    import msgpack
    lib = lib or os.path.join(ROOT, "toppra_amd", "libtoppra_hip.so")
    out = {}
    for elf in code_objects(lib):
        for desc in _notes(elf):
            meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
            for k in meta.get("amdhsa.kernels", []):
                out[k[".name"]] = dict(vgpr=k.get(".vgpr_count", 0), agpr=k.get(".agpr_count", 0),
                                       scratch=k.get(".private_segment_fixed_size", 0), lds=k.get(".group_segment_fixed_size", 0),
                                       sgpr_spill=k.get(".sgpr_spill_count", 0), vgpr_spill=k.get(".vgpr_spill_count", 0))
    return out


def demangle(names):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            txt = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True).stdout
            if len(txt.splitlines()) == len(names):
                return dict(zip(names, txt.splitlines()))
        except OSError:
            pass
    return {n: n for n in names}


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else None
    pat = [a for a in sys.argv[1:] if not a.endswith(".so")]
    ks = kernels(lib)
    dm = demangle(sorted(ks))
    for n in sorted(ks, key=lambda n: dm[n]):
        if pat and not any(p in dm[n] for p in pat):
            continue
        r = ks[n]
        print("%-110s vgpr %3d agpr %3d scratch %5d B lds %6d B sgpr spills %3d" % (dm[n][:110], r["vgpr"], r["agpr"], r["scratch"], r["lds"], r["sgpr_spill"]))
