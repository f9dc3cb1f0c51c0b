#!/usr/bin/env python3
"""Static check of every GPU kernel inside detr_amd/libdetr_match.so (no GPU needed).

    python tools/check_code_objects.py [path/to/lib.so]

Reads the kernel descriptors of the gfx950 code objects bundled in the library and fails when a kernel
  * uses private (scratch) memory: register spills or a per-thread array the compiler could not keep in
    registers -- every access is a memory round trip;
  * asks for the dispatch-packet pointer: the compiler then reads the workgroup size from the AQL packet,
    which lives in the queue's ring buffer in HOST memory (it does so to address a per-thread array it has
    moved to LDS).  One such read per wavefront cost 14 us per launch in cost_small_kernel<3, 1> before the
    class costs were held as a vector value instead of an array.
Prints one line per kernel: VGPRs, LDS bytes, scratch bytes.
"""
import os
import shutil
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sections_and_symbols(blob):
    """Minimal ELF64 little-endian reader: {section index: (addr, offset, size)}, [(name, value, size, shndx)]."""
    assert blob[:4] == b"\x7fELF" and blob[4] == 2 and blob[5] == 1
    shoff, = struct.unpack_from("<Q", blob, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
    sections = []
    for i in range(shnum):
        name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from("<IIQQQQIIQQ", blob,
                                                                                           shoff + i * shentsize)
        sections.append(dict(type=typ, addr=addr, off=off, size=size, link=link, entsize=entsize))
    symbols = []
    for sec in sections:
        if sec["type"] not in (2, 11):          # SHT_SYMTAB, SHT_DYNSYM
            continue
        strtab = sections[sec["link"]]
        for k in range(sec["size"] // sec["entsize"]):
            st_name, st_info, st_other, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", blob,
                                                                                         sec["off"] + k * sec["entsize"])
            end = blob.index(b"\0", strtab["off"] + st_name)
            symbols.append((blob[strtab["off"] + st_name:end].decode(), st_value, st_size, st_shndx))
    return sections, symbols


def kernels_of(code_object):
    """[(kernel name, vgprs-granule field, lds bytes, scratch bytes, wants dispatch ptr)] from the 64-byte
    kernel descriptors (symbols ending in .kd)."""
    blob = open(code_object, "rb").read()
    sections, symbols = sections_and_symbols(blob)
    out, seen = [], set()
    for name, value, size, shndx in symbols:
        if not name.endswith(".kd") or name in seen or shndx == 0 or shndx >= len(sections):
            continue
        seen.add(name)
        sec = sections[shndx]
        at = sec["off"] + (value - sec["addr"])
        lds, scratch = struct.unpack_from("<II", blob, at)
        rsrc1, = struct.unpack_from("<I", blob, at + 48)
        props, = struct.unpack_from("<H", blob, at + 56)
        vgpr_granules = rsrc1 & 0x3F           # (granulated_workitem_vgpr_count + 1) * 8 registers on gfx90a+
        out.append((name[:-3], (vgpr_granules + 1) * 8, lds, scratch, bool(props & 0x2)))
    return out


def demangle(names):
    for tool in (os.path.join(LLVM, "llvm-cxxfilt"), "c++filt"):
        try:
            res = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
            if res.returncode == 0:
                return res.stdout.split("\n")[:len(names)]
        except OSError:
            pass
    return names


def check(lib):
    """Returns (lines, problems) for the library's kernels."""
    work = tempfile.mkdtemp(prefix="dm_co_")
    try:
        copy = os.path.join(work, os.path.basename(lib))
        shutil.copy(lib, copy)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], check=True, capture_output=True)
        objects = sorted(f for f in os.listdir(work) if "amdgcn" in f)
        if not objects:
            raise RuntimeError("no gfx code objects found in " + lib)
        lines, problems = [], []
        for f in objects:
            ks = kernels_of(os.path.join(work, f))
            for (name, vgprs, lds, scratch, dispatch), pretty in zip(ks, demangle([k[0] for k in ks])):
                short = pretty[:110]
                lines.append(f"{short:110s}  vgpr<={vgprs:3d}  lds={lds:6d}  scratch={scratch}")
                if scratch:
                    problems.append(f"{short}: {scratch} bytes of private memory per thread")
                if dispatch:
                    problems.append(f"{short}: reads the dispatch packet (host memory)")
        return lines, problems
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "detr_amd", "libdetr_match.so")
    lines, problems = check(lib)
    print("\n".join(lines))
    print(f"{len(lines)} kernels")
    for p in problems:
        print("PROBLEM:", p)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
