"""Per-kernel resource table of libgravitas_hip.so (VGPRs, SGPRs, LDS, scratch, waves/SIMD), decoded
from the AMDGPU metadata notes of the embedded gfx950 code objects.  No GPU needed.
    python tools/kernel_resources.py > profiles/r01_kernel_resources.txt"""
import os
import re
import struct
import sys

import msgpack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernels(path):
    data = open(path, "rb").read()
    out, pos = {}, 0
    while True:
        base = data.find(MAGIC, pos)
        if base < 0:
            return out
        pos = base + 1
        (n,) = struct.unpack_from("<Q", data, base + 24)
        off = base + 32
        for _ in range(n):
            o, s, t = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + t].decode()
            off += t
            if "amdgcn" not in triple or s == 0:
                continue
            elf = data[base + o:base + o + s]
            (shoff,) = struct.unpack_from("<Q", elf, 0x28)
            shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
            for k in range(shnum):
                sh = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + k * shentsize)
                if sh[1] != 7:
                    continue
                p, end = sh[4], sh[4] + sh[5]
                while p < end:
                    namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
                    p += 12
                    name = elf[p:p + namesz]
                    p += (namesz + 3) & ~3
                    desc = elf[p:p + descsz]
                    p += (descsz + 3) & ~3
                    if name.startswith(b"AMDGPU") and ntype == 32:
                        for kd in msgpack.unpackb(desc, raw=False, strict_map_key=False)["amdhsa.kernels"]:
                            out[kd[".name"]] = kd


def code_objects(path):
    """The embedded gfx950 ELF images of a bundle-carrying host library."""
    data = open(path, "rb").read()
    pos = 0
    while True:
        base = data.find(MAGIC, pos)
        if base < 0:
            return
        pos = base + 1
        (n,) = struct.unpack_from("<Q", data, base + 24)
        off = base + 32
        for _ in range(n):
            o, s, t = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + t].decode()
            off += t
            if "amdgcn" in triple and s:
                yield data[base + o:base + o + s]


def kernel_code_hash(path, pretty_name):
    """sha256 (first 16 hex digits) of one kernel's machine code + kernel descriptor, located
    through the code object's symbol table.  `pretty_name` as printed by this tool, e.g.
    'integrate_segment_kernel<1,1,0>'.  Changes whenever the compiled kernel changes and only then
    (the descriptor's offset to the code is masked: it depends on the other kernels of the unit):
    profiles/traffic.json is stamped with it and bench.py drops the committed PMC figures when
    the stamp and the library on disk disagree."""
    import hashlib
    for elf in code_objects(path):
        (shoff,) = struct.unpack_from("<Q", elf, 0x28)
        shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
        secs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + k * shentsize) for k in range(shnum)]
        for sh in secs:
            if sh[1] not in (2, 11):  # SHT_SYMTAB, SHT_DYNSYM
                continue
            strtab = secs[sh[6]]
            h = None
            parts = {}
            for k in range(sh[5] // 24):
                st_name, st_info, _o, st_shndx, st_value, st_size = struct.unpack_from(
                    "<IBBHQQ", elf, sh[4] + k * 24)
                end = elf.index(b"\0", strtab[4] + st_name)
                name = elf[strtab[4] + st_name:end].decode()
                base = name[:-3] if name.endswith(".kd") else name
                if pretty(base) != pretty_name or st_shndx == 0 or st_shndx >= len(secs):
                    continue
                sec = secs[st_shndx]
                o = st_value - sec[3] + sec[4]
                parts["kd" if name.endswith(".kd") else "text"] = elf[o:o + st_size]
            if "text" in parts:
                # the descriptor's KERNEL_CODE_ENTRY_BYTE_OFFSET (bytes 16-23: the distance from the
                # descriptor to the code) moves whenever ANOTHER kernel of the translation unit changes
                # size; it is masked so that the stamp follows this kernel's code and resources only
                kd = bytearray(parts.get("kd", b""))
                if len(kd) >= 24:
                    kd[16:24] = b"\0" * 8
                h = hashlib.sha256(parts["text"] + bytes(kd)).hexdigest()[:16]
                return h
    return None


def pretty(mangled):
    m = re.search(r"\d+([a-z_0-9]+_kernel)(I[A-Za-z0-9]*E)?", mangled)
    if not m:
        return mangled[:48]
    targs = re.findall(r"Li(\d+)E", m.group(2) or "")
    return m.group(1) + ("<" + ",".join(targs) + ">" if targs else "")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--hashes"]
    path = args[0] if args else os.path.join(ROOT, "blackhole-simulation_amd", "libgravitas_hip.so")
    ks = kernels(path)
    if "--hashes" in sys.argv:  # {pretty name: code hash} of every kernel, for profiles/traffic.json
        import json
        names = sorted({pretty(n) for n in ks})
        print(json.dumps({n: kernel_code_hash(path, n) for n in names}, indent=1))
        sys.exit(0)
    print("# %s: %d kernels, gfx950, wave64.  waves/SIMD = floor(512 / ceil8(vgpr + agpr)), capped at 8" %
          (os.path.basename(path), len(ks)))
    print("# template arguments: <metric kind (0 BL, 1 KS, 2 Schwarzschild), arith (0 STRICT, 1 FAST), method (0 RKF45, 1 RK4, 2 symplectic)>"
          " or <arith>")
    print("%-44s %5s %5s %5s %7s %8s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "lds_B", "scratch", "waves"))
    for n, kd in sorted(ks.items(), key=lambda kv: pretty(kv[0])):
        v = kd[".vgpr_count"] + kd.get(".agpr_count", 0)
        waves = min(8, 512 // max(8, (v + 7) // 8 * 8))
        print("%-44s %5d %5d %5d %7d %8d %6d" % (pretty(n), kd[".vgpr_count"], kd.get(".agpr_count", 0), kd[".sgpr_count"],
                                                 kd[".group_segment_fixed_size"], kd[".private_segment_fixed_size"], waves))
