#!/usr/bin/env python
"""Instruction histogram of a kernel's hot loop, priced with measured issue costs.  No GPU needed for
the histogram; the costs come from tools/valu_microbench (run on the GPU box).

    python tools/isa_histogram.py 'wgsl_symplectic_pk_kernel' [--costs profiles/r03_valu_costs.json]
                                  [--dynamic-valu N_per_unit] [--units-per-iter K]

The library's embedded gfx950 code object is disassembled with llvm-objdump; the kernel's largest
backward-branch loop (the march / try loop) is taken as the hot loop and its instructions are counted
by mnemonic and grouped by class.  With --costs every VALU mnemonic is priced in cycles per wave64
instruction on one SIMD-32 (measured; a mnemonic the microbenchmark does not cover takes the cost of
its class representative), giving the issue cycles of one static pass through the loop, the mean
cost per VALU instruction and -- for a kernel whose dynamic VALU count is known from SQ_INSTS_VALU
-- the VALU-issue floor per unit of work."""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter, OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as kr  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

# class of a mnemonic; first match wins.  The representative is the microbenchmark case whose cost an
# uncovered mnemonic of the class takes.
CLASSES = [
    ("f64 transcendental (v_rcp/rsq/sqrt_f64)", r"^v_(rcp|rsq|sqrt)_f64", "v_rcp_f64"),
    ("f64 divide helpers (v_div_scale/fmas/fixup)", r"^v_div_(scale|fmas|fixup)_f64", "v_div_fmas_f64"),
    ("f64 fma / mul / add / min / max", r"^v_(fma|fmac|mul|add|max|min)_f64", "v_fma_f64"),
    ("f64 compare", r"^v_cmpx?_\w+_f64", "v_cmp_f64"),
    ("f64 other (ldexp, frexp, fract, round, trig_preop)", r"^v_\w+_f64$", "v_ldexp_f64"),
    ("conversion with an f64 side", r"^v_cvt_(f64_\w+|\w+_f64)", "v_cvt_f64_f32"),
    ("packed f32 (v_pk_*)", r"^v_pk_", "v_pk_fma_f32"),
    ("f32 transcendental (v_rcp/rsq/sqrt/exp/log/sin/cos_f32)", r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)(_iflag)?_f32", "v_rcp_f32"),
    ("f32 fma / mul / add / sub / min / max / med3", r"^v_(fma|fmac|mac|mad|mul|add|sub|subrev|max|min|med3|max3|min3|fmaak|fmamk)_(legacy_)?f32", "v_fma_f32"),
    ("f32 compare", r"^v_cmpx?_\w+_f32", "v_cmp_f32"),
    ("f32 other (fract, floor, rndne, ldexp, frexp)", r"^v_(fract|floor|ceil|trunc|rndne|ldexp|frexp_mant|frexp_exp_i32)_f32", "v_fract_f32"),
    ("conversion f32 <-> int / f16", r"^v_cvt_", "v_cvt_i32_f32"),
    ("lane access (v_readlane / v_writelane / v_readfirstlane)", r"^v_(readlane|writelane|readfirstlane)_b32", "v_readlane"),
    ("64-bit move", r"^v_mov_b64", "v_mov_b64"),
    ("select (v_cndmask)", r"^v_cndmask_b32", "v_cndmask_b32"),
    ("integer multiply", r"^v_mul_(lo|hi)_[ui]32|^v_mad_[ui]64", "v_mul_lo_u32"),
    ("integer / logic / shift / compare / 32-bit move", r"^v_", "v_add_u32"),
    ("scalar ALU / branch", r"^s_(?!waitcnt|nop|load|buffer|store|sleep|setprio)", None),
    ("scalar memory", r"^s_(load|buffer_load|store)", None),
    ("wait / nop", r"^s_(waitcnt|nop|sleep|setprio)", None),
    ("vector memory (global / flat / scratch / buffer)", r"^(global|flat|scratch|buffer)_", None),
    ("LDS", r"^ds_", None),
]


def disassemble(lib, pretty_name):
    """[(address, mnemonic, operands)] of the kernel whose pretty name matches."""
    for elf in kr.code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(elf)
            path = f.name
        try:
            txt = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", "--no-show-raw-insn", path], capture_output=True,
                                 text=True, check=True).stdout
        finally:
            os.unlink(path)
        cur, out = None, []
        for line in txt.splitlines():
            m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
            if m:
                if cur and out:
                    return out
                cur = m.group(2) if kr.pretty(m.group(2)) == pretty_name else None
                continue
            if cur:
                m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
                if m:
                    out.append((int(m.group(3), 16), m.group(1), m.group(2)))
        if cur and out:
            return out
    return None


def branch_target(addr, mnem, ops, insts_by_addr):
    if not mnem.startswith("s_cbranch") and mnem != "s_branch":
        return None
    m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", ops)  # objdump prints <kernel+0xOFF>
    if m:
        return ("off", int(m.group(1), 16))
    m = re.match(r"^(-?\d+)$", ops.strip())
    if m:  # raw simm16 (printed unsigned): target = addr + 4 + simm16 * 4
        v = int(m.group(1))
        if v >= 32768:
            v -= 65536
        return ("abs", addr + 4 + v * 4)
    return None


def hot_loop(insts):
    """Largest backward-branch loop: (start index, end index inclusive)."""
    base = insts[0][0]
    idx = {a: k for k, (a, _, _) in enumerate(insts)}
    best = None
    for k, (a, mn, ops) in enumerate(insts):
        t = branch_target(a, mn, ops, idx)
        if not t:
            continue
        ta = base + t[1] if t[0] == "off" else t[1]
        if ta in idx and idx[ta] <= k:
            span = k - idx[ta]
            if best is None or span > best[1] - best[0]:
                best = (idx[ta], k)
    return best


def classify(mnem):
    m = re.sub(r"_(e32|e64|sdwa|dpp|vi|gfx\d+)$", "", mnem)
    for name, pat, rep in CLASSES:
        if re.search(pat, m):
            return name, rep, m
    return "other", None, m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel")
    ap.add_argument("--lib", default=os.path.join(ROOT, "blackhole-simulation_amd", "libgravitas_hip.so"))
    ap.add_argument("--costs", default=None, help="JSON written by tools/valu_microbench")
    ap.add_argument("--dynamic-valu", type=float, default=None,
                    help="measured wave-level VALU instructions per unit of work (SQ_INSTS_VALU / units)")
    ap.add_argument("--unit", default="wave-try")
    ap.add_argument("--whole-kernel", action="store_true", help="histogram of the whole kernel, not the hot loop")
    ap.add_argument("--loop", default=None, help="lo:hi instruction indices of the loop to count (default: largest backward branch)")
    ap.add_argument("--json", action="store_true")
    args = ap.parse_args()
    insts = disassemble(args.lib, args.kernel)
    if not insts:
        raise SystemExit("kernel %s not found in %s" % (args.kernel, args.lib))
    if args.loop:
        lo, hi = (int(x) for x in args.loop.split(":"))
    else:
        lo, hi = (0, len(insts) - 1) if args.whole_kernel else (hot_loop(insts) or (0, len(insts) - 1))
    body = insts[lo:hi + 1]
    costs = None
    if args.costs:
        with open(args.costs) as f:
            costs = json.load(f)["cycles"]
    per_class = OrderedDict((c[0], Counter()) for c in CLASSES)
    per_class["other"] = Counter()
    rep_of = {c[0]: c[2] for c in CLASSES}
    for _, mn, _ in body:
        cname, _, base = classify(mn)
        per_class[cname][base] += 1
    rows, valu_n, valu_cyc = [], 0, 0.0
    for cname, cnt in per_class.items():
        n = sum(cnt.values())
        if not n:
            continue
        is_valu = rep_of.get(cname) is not None
        cyc = None
        if is_valu and costs:
            cyc = 0.0
            for mn, k in cnt.items():
                key = mn if mn in costs else None
                if key is None:  # e.g. v_fmac_f64 -> v_fma_f64, v_cmp_gt_f64 -> v_cmp_f64
                    key = rep_of[cname]
                cyc += k * costs[key]
        if is_valu:
            valu_n += n
            valu_cyc += cyc or 0.0
        rows.append({"class": cname, "instructions": n, "issue_cycles": None if cyc is None else round(cyc, 1),
                     "mnemonics": dict(cnt.most_common())})
    res = {"kernel": args.kernel, "code_hash": kr.kernel_code_hash(args.lib, args.kernel),
           "scope": "whole kernel" if args.whole_kernel else "hot loop (largest backward branch)",
           "loop_instructions": len(body), "kernel_instructions": len(insts), "valu_instructions_static": valu_n,
           "classes": rows}
    if costs:
        res["valu_issue_cycles_static_pass"] = round(valu_cyc, 1)
        res["mean_cycles_per_valu_instruction"] = round(valu_cyc / max(valu_n, 1), 3)
        if args.dynamic_valu:
            # dynamic count from the PMC pass, mean cost from the static mix of the loop
            res["dynamic_valu_per_" + args.unit] = args.dynamic_valu
            res["valu_issue_cycles_per_" + args.unit] = round(args.dynamic_valu * valu_cyc / max(valu_n, 1), 1)
    if args.json:
        print(json.dumps(res, indent=1))
        return
    print("# %s  (code object %s)  %s: %d instructions of %d" %
          (args.kernel, res["code_hash"], res["scope"], len(body), len(insts)))
    print("%-62s %6s %10s" % ("class", "count", "cycles"))
    for r in rows:
        print("%-62s %6d %10s" % (r["class"], r["instructions"], "" if r["issue_cycles"] is None else "%.1f" % r["issue_cycles"]))
        top = ", ".join("%s x%d" % kv for kv in list(r["mnemonics"].items())[:8])
        print("    " + top)
    if costs:
        print("VALU: %d instructions, %.1f issue cycles per static pass, %.3f cycles per instruction on average" %
              (valu_n, valu_cyc, valu_cyc / max(valu_n, 1)))
        if args.dynamic_valu:
            print("dynamic: %.1f VALU per %s (SQ_INSTS_VALU) -> %.1f issue cycles per %s at the loop's mean cost" %
                  (args.dynamic_valu, args.unit, res["valu_issue_cycles_per_" + args.unit], args.unit))


if __name__ == "__main__":
    main()
