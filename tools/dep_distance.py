#!/usr/bin/env python3
"""How far behind its nearest producer does every VALU instruction of a kernel's hot loop sit?  Histogram of the
distance (in VALU instructions) between an instruction and the latest earlier instruction that wrote one of its
source registers -- 1 = it reads the result of the instruction just before it (served by the forwarding path),
8 = eight or more (or produced outside the loop).  Used to compare two schedules of the same instructions
(profiles/EXPERIMENTS.md N: the FAST f64 try loop with and without the post-RA machine scheduler).
usage: tools/dep_distance.py 'integrate_segment_kernel<1,1,0>' lib_a.so [lib_b.so ...]"""
import collections
import re
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_histogram as ih  # noqa: E402


def regs(tok):
    out = set()
    for m in re.finditer(r"\b([vs])\[(\d+):(\d+)\]|\b([vs])(\d+)\b", tok):
        if m.group(1):
            out |= {m.group(1) + str(i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            out.add(m.group(4) + m.group(5))
    if "vcc" in tok:
        out.add("vcc")
    return out


def analyse(lib, name):
    ins = ih.disassemble(lib, name)
    lo, hi = ih.hot_loop(ins)
    last, hist, n = {}, collections.Counter(), 0
    for i in range(lo, hi + 1):
        _, m, o = ins[i]
        if not m.startswith("v_"):
            continue
        n += 1
        ops = [x.strip() for x in o.split(",")]
        dst = regs(ops[0]) if ops else set()
        srcs = set()
        for x in ops[1:]:
            srcs |= regs(x)
        if m.startswith(("v_fmac", "v_mac")):
            srcs |= dst
        d = min([n - last[r] for r in srcs if r in last] or [99])
        hist[min(d, 8)] += 1
        for r in dst:
            last[r] = n
    print("%-28s loop %d..%d  VALU %d  distance to producer: %s" % (os.path.basename(lib), lo, hi, n, dict(sorted(hist.items()))))


if __name__ == "__main__":
    for lib in sys.argv[2:]:
        analyse(lib, sys.argv[1])
