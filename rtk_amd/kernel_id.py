"""Identity of the traversal kernels as they sit in the built library: a hash over the MACHINE CODE of named kernels.

bench.py ties the committed rocprofv3 counter summaries (profiles/rNN_*_pmc.json) to the kernels they were measured on.
Hashing source files made that guard fire on comments and host-side edits (round 2); this hashes the gfx950 code of the
kernels themselves: every AMDGPU ELF image inside librtk_amd.so is parsed (the HIP fat binary of each translation unit,
and the hand-assembled code object of rtk_packet_hot that the library carries as a byte array), and the bytes of the
requested function symbols plus their kernel descriptors go into the hash. Pure Python, no tools, no GPU.
"""
import hashlib
import os
import struct

EM_AMDGPU = 224


def _elf_images(blob):
    """(offset, size) of every 64-bit little-endian AMDGPU ELF image embedded in `blob`."""
    out = []
    pos = 0
    while True:
        i = blob.find(b"\x7fELF\x02\x01", pos)
        if i < 0:
            break
        pos = i + 4
        if i + 64 > len(blob):
            continue
        e_machine = struct.unpack_from("<H", blob, i + 18)[0]
        if e_machine != EM_AMDGPU:
            continue
        e_shoff, = struct.unpack_from("<Q", blob, i + 40)
        e_shentsize, e_shnum = struct.unpack_from("<HH", blob, i + 58)
        size = e_shoff + e_shentsize * e_shnum
        if size <= 64 or i + size > len(blob):
            continue
        out.append((i, size))
    return out


def _symbols(img):
    """name -> (bytes of the symbol) for FUNC and OBJECT symbols of one ELF image."""
    e_shoff, = struct.unpack_from("<Q", img, 40)
    e_shentsize, e_shnum, e_shstrndx = struct.unpack_from("<HHH", img, 58)
    secs = []
    for k in range(e_shnum):
        sh = struct.unpack_from("<IIQQQQIIQQ", img, e_shoff + k * e_shentsize)
        secs.append(dict(name=sh[0], type=sh[1], addr=sh[3], offset=sh[4], size=sh[5], link=sh[6], entsize=sh[9]))
    out = {}
    for s in secs:
        if s["type"] not in (2, 11):           # SHT_SYMTAB, SHT_DYNSYM
            continue
        strtab = secs[s["link"]]
        for k in range(s["size"] // 24):
            st_name, st_info, _, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", img, s["offset"] + 24 * k)
            if (st_info & 15) not in (1, 2) or st_size == 0 or st_shndx == 0 or st_shndx >= len(secs):
                continue
            end = img.find(b"\0", strtab["offset"] + st_name)
            name = img[strtab["offset"] + st_name:end].decode("ascii", "replace")
            sec = secs[st_shndx]
            if sec["type"] == 8:                # SHT_NOBITS
                continue
            off = sec["offset"] + (st_value - sec["addr"])
            out[name] = img[off:off + st_size]
    return out


def kernel_code_sha16(lib_path, name_parts):
    """sha256 (16 hex digits) over the code and kernel descriptors of every kernel in `lib_path` whose (mangled) symbol name
    contains ALL of the strings of one entry of `name_parts` (a list of tuples). None if the library or a kernel is missing."""
    if not os.path.exists(lib_path):
        return None
    blob = open(lib_path, "rb").read()
    syms = {}
    for off, size in _elf_images(blob):
        try:
            syms.update(_symbols(blob[off:off + size]))
        except Exception:
            continue
    h = hashlib.sha256()
    for parts in name_parts:
        names = sorted(n for n in syms if all(p in n for p in parts))
        if not names:
            return None
        for n in names:
            h.update(n.encode())
            h.update(syms[n])
    return h.hexdigest()[:16]


# what each bench workload's committed counter summary was measured on
WORKLOAD_KERNELS = {
    # the hand-written packet kernel (rtk_packet_beam2.S: two tiles per wave) and the C++ packet kernel behind it (the tiles it hands back)
    "coherent": [("rtk_packet_beam2",), ("rtk_trace_packet_kernelILb0E",)],
    # the hand-written per-lane kernels and rtk_trace_kernel<MODE 0 / 1, COUNT false, FILT false, QN true> behind them (the rays they hand back)
    "incoherent": [("rtk_lane_hot_closest",), ("rtk_trace_kernelILi0ELb0ELb0ELb1E",)],
    "shadow": [("rtk_lane_hot_any",), ("rtk_trace_kernelILi1ELb0ELb0ELb1E",)],
}


def workload_kernel_sha16(lib_path, workload):
    return kernel_code_sha16(lib_path, WORKLOAD_KERNELS[workload])


if __name__ == "__main__":
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "librtk_amd.so")
    for w in WORKLOAD_KERNELS:
        print(w, workload_kernel_sha16(lib, w))
