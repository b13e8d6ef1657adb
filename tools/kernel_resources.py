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


def pretty(mangled):
    m = re.search(r"\d+([a-z_0-9]+_kernel)(I[A-Za-z0-9]*E)?", mangled)
    if not m:
        return mangled[:48]
    targs = re.findall(r"Li(\d+)E", m.group(2) or "")
    return m.group(1) + ("<" + ",".join(targs) + ">" if targs else "")


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "blackhole-simulation_amd", "libgravitas_hip.so")
    ks = kernels(path)
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
