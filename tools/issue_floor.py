#!/usr/bin/env python
"""VALU issue accounting of the dominant kernels from hardware counters (no GPU needed: reads the
rocprofv3 databases tools/profile_gpu.sh wrote).

    python tools/issue_floor.py gpurun_out/prof_r03d profiles r03

Two things come out (profiles/<tag>_valu_issue.json + .txt):

1. The issue model of a gfx950 SIMD, CALIBRATED on tools/valu_microbench (one kernel per mnemonic,
   run under the same counters): what one wave64 instruction of each kind adds to
   SQ_ACTIVE_INST_VALU (quad-cycles the VALU is held), how often two instructions share a quad-cycle
   (SQ_ACTIVE_INST_VALU2), which SQ_INSTS_VALU_* class counter it lands in, and its cost in shader
   cycles measured with s_memtime.  Result: every VALU instruction holds the SIMD for one quad-cycle
   (4 cycles) -- f64, packed f32, conversions, compares, selects and most integer ops included --
   except transcendentals (2 quad-cycles f32, 4 quad-cycles f64), and the plain full-rate 32-bit ops
   (v_fma/mul/add_f32, v_mov_b32, v_add_u32, and/or/xor, lshr) can pair up two to a quad-cycle.
2. For each dominant kernel: the dynamic instruction mix by class per ray-step, the quad-cycles it
   holds the VALU, and the issue occupancy
       (SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2) x 4 cycles / (SIMDs x elapsed cycles)
   -- the fraction of the chip's VALU issue slots the kernel fills.  SQ_ACTIVE_INST_VALU alone
   over-counts paired instructions (it gives 1.16 for the one-ray f32 march); SQ_INSTS_VALU x 2 cycles
   under-counts everything that is not a full-rate f32 op (0.44 for the packed march).
"""
import collections
import json
import os
import sqlite3
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
DST = sys.argv[2] if len(sys.argv) > 2 else "profiles"
TAG = sys.argv[3] if len(sys.argv) > 3 else "r03"
N_SIMD = 1024

CLASS_COUNTERS = ["SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64",
                  "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_TRANS_F32",
                  "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64"]
# quad-cycles one instruction of the class holds the VALU (calibrated below; asserted against the calibration)
QUADS = {"SQ_INSTS_VALU_TRANS_F64": 4.0, "SQ_INSTS_VALU_TRANS_F32": 2.0}

KERNELS = [  # (pretty, needle in the trace, pass suffix, unit count key in the bench line)
    ("integrate_segment_kernel<1,1,0>", "integrate_segment_kernel<1, 1, 0>", "", "FAST f64 RKF45 (the bench line, configs[2])"),
    ("integrate_segment_kernel<1,0,0>", "integrate_segment_kernel<1, 0, 0>", "_strict", "STRICT f64 RKF45 (reference order)"),
    ("wgsl_symplectic_pk_kernel", "wgsl_symplectic_pk_kernel", "_c4", "f32 compute march, two rays per lane (configs[3])"),
    ("wgsl_symplectic_fast_kernel", "wgsl_symplectic_fast_kernel", "_c4fast", "f32 compute march, one ray per lane"),
]


def counters(db, needle=None):
    """{kernel: {counter: average per dispatch}} (or the one kernel matching `needle`)."""
    if not os.path.exists(db):
        return {}
    rows = sqlite3.connect(db).execute(
        "select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
    out = collections.defaultdict(dict)
    for k, c, a in rows:
        out[k][c] = a
    if needle is None:
        return out
    for k, v in out.items():
        if needle in k:
            return v
    return {}


def merged(labels, needle=None):
    d = {} if needle else collections.defaultdict(dict)
    for lab in labels:
        c = counters(os.path.join(SRC, lab, "bench_results.db"), needle)
        if needle:
            for k, v in c.items():
                d.setdefault(k, v)
        else:
            for k, v in c.items():
                for kk, vv in v.items():
                    d[k].setdefault(kk, vv)
    return d


def calibration():
    mb = merged(["mb_occ", "mb_cls32", "mb_cls64", "mb_sq"])
    costs = {}
    for w in (8, 4, 2, 1):
        p = os.path.join(SRC, "valu_costs_w%d.json" % w)
        if os.path.exists(p):
            costs[w] = json.load(open(p))
    rows = []
    for kname, c in sorted(mb.items()):
        name = kname.split("(")[0]
        if not name.startswith("k_"):
            continue
        name = name[2:]
        n = c.get("SQ_INSTS_VALU")
        if not n:
            continue
        per_stmt = 2 if name.startswith("mix_") else 1
        stmts = n / per_stmt
        row = {"case": name, "instructions_per_statement": per_stmt,
               "active_quads_per_statement": round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / stmts, 3),
               "paired_quads_per_statement": round(c.get("SQ_ACTIVE_INST_VALU2", 0.0) / stmts, 3),
               "class": [k.replace("SQ_INSTS_VALU_", "") for k in CLASS_COUNTERS if c.get(k, 0.0) / stmts > 0.5]}
        row["issue_quads_per_statement"] = round(row["active_quads_per_statement"] - row["paired_quads_per_statement"], 3)
        for fk in ("SQ_INSTS_VALU_FLOPS_FP32", "SQ_INSTS_VALU_FLOPS_FP64", "SQ_INSTS_VALU_FLOPS_FP32_TRANS",
                   "SQ_INSTS_VALU_FLOPS_FP64_TRANS"):
            if c.get(fk):
                row.setdefault("flops_counted_per_statement", {})[fk.replace("SQ_INSTS_VALU_FLOPS_", "")] = round(c[fk] / stmts, 2)
        for w, cj in costs.items():
            key = "v_" + name
            v = cj.get("cycles", {}).get(key, cj.get("pairs_cycles_per_statement", {}).get(name))
            if v is not None:
                row["cycles_s_memtime_w%d" % w] = v
        rows.append(row)
    return rows, costs


def kernel_entry(pretty, needle, sfx, what):
    c = merged(["pmc_occ" + sfx, "pmc_cls32" + sfx, "pmc_cls64" + sfx, "pmc_sq" + sfx], needle)
    if not c.get("SQ_INSTS_VALU"):
        return None
    bj = os.path.join(SRC, "pmc_cls32%s_bench.json" % sfx)
    steps = rays = None
    if os.path.exists(bj):
        try:
            cfg = json.load(open(bj))["config"]
            steps, rays = cfg.get("accepted_steps_per_frame"), cfg.get("rays")
        except Exception:
            pass
    tot = c["SQ_INSTS_VALU"]
    ent = {"kernel": pretty, "what": what, "rays": rays, "accepted_steps_per_launch": steps,
           "valu_wave_instructions_per_launch": tot}
    if steps:
        ent["valu_lane_instructions_per_ray_step"] = round(tot * 64.0 / steps, 1)
    mix, classed, quads_model = collections.OrderedDict(), 0.0, 0.0
    for k in CLASS_COUNTERS:
        v = c.get(k, 0.0)
        classed += v
        q = QUADS.get(k, 1.0)
        quads_model += v * q
        if v:
            mix[k.replace("SQ_INSTS_VALU_", "")] = {"share": round(v / tot, 4), "quad_cycles_each": q,
                                                   "per_ray_step": round(v * 64.0 / steps, 2) if steps else None}
    other = tot - classed
    quads_model += other
    mix["other (moves, selects, compares, min/max, ldexp/frexp/round, lane access, bit ops)"] = {
        "share": round(other / tot, 4), "quad_cycles_each": 1.0,
        "per_ray_step": round(other * 64.0 / steps, 2) if steps else None}
    ent["mix"] = mix
    arith = sum(c.get(k, 0.0) * QUADS.get(k, 1.0) for k in CLASS_COUNTERS[:8])
    act, act2, gui = c.get("SQ_ACTIVE_INST_VALU"), c.get("SQ_ACTIVE_INST_VALU2"), c.get("GRBM_GUI_ACTIVE")
    ent["quad_cycles_model_per_launch"] = quads_model
    ent["arithmetic_share_of_quad_cycles"] = round(arith / quads_model, 4)
    if act and act2 is not None and gui:
        cyc = gui / 8.0
        ent.update({
            "sq_active_inst_valu": act, "sq_active_inst_valu2": act2, "elapsed_cycles_per_xcd": cyc,
            "model_vs_counter": round(quads_model / act, 4),
            "paired_fraction_of_instructions": round(2.0 * act2 / tot, 4),
            "valu_issue_occupancy": round((act - act2) * 4.0 / N_SIMD / cyc, 4),
            "naive_active_inst_valu_frac": round(act * 4.0 / N_SIMD / cyc, 4),
            "naive_insts_x2cycles_frac": round(tot * 2.0 / N_SIMD / cyc, 4),
            "naive_insts_x4cycles_frac": round(tot * 4.0 / N_SIMD / cyc, 4)})
        if steps:
            # SIMD cycles one ray-step costs (a wave instruction serves 64 rays): 1024 SIMDs x clock / this = ray-steps/s
            ent["simd_issue_cycles_per_ray_step"] = round((act - act2) * 4.0 / steps, 2)
            ent["simd_arithmetic_cycles_per_ray_step"] = round(arith * 4.0 / steps, 2)
    fl = {k: c.get("SQ_INSTS_VALU_FLOPS_" + k) for k in ("FP32", "FP64", "FP32_TRANS", "FP64_TRANS")}
    if any(fl.values()):
        ent["flops_counted_per_launch"] = fl
        if steps:
            ent["flops_counted_per_ray_step"] = {k: round(v / steps, 1) for k, v in fl.items() if v}
    for k in ("SQ_INSTS_SALU", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES"):
        if c.get(k) is not None:
            ent[k.lower()] = c[k]
    return ent


def main():
    os.makedirs(DST, exist_ok=True)
    cal, costs = calibration()
    kern = [e for e in (kernel_entry(*k) for k in KERNELS) if e]
    hashes = {}
    hp = os.path.join(SRC, "code_hashes.json")
    if os.path.exists(hp):
        hashes = json.load(open(hp))
    for e in kern:
        e["code_hash"] = hashes.get(e["kernel"])
    out = {"source": SRC, "simds": N_SIMD,
           "occupancy_definition": "(SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2) * 4 cycles / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)",
           "kernels": kern, "calibration": cal,
           "microbench_method": {w: c.get("method") for w, c in costs.items()}}
    json.dump(out, open(os.path.join(DST, "%s_valu_issue.json" % TAG), "w"), indent=1)

    L = ["# VALU issue accounting (tools/issue_floor.py over %s)" % SRC, "",
         "## dominant kernels", "",
         "occupancy = (SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2) x 4 cycles / (1024 SIMDs x elapsed cycles)", ""]
    for e in kern:
        L.append("%s  -- %s" % (e["kernel"], e["what"]))
        if "valu_issue_occupancy" in e:
            L.append("  VALU issue occupancy %.4f   (naive: ACTIVE_INST_VALU alone %.4f, INSTS x 2 cycles %.4f, INSTS x 4 cycles %.4f)"
                     % (e["valu_issue_occupancy"], e["naive_active_inst_valu_frac"], e["naive_insts_x2cycles_frac"],
                        e["naive_insts_x4cycles_frac"]))
            L.append("  instructions sharing a quad-cycle: %.1f %%; class model / SQ_ACTIVE_INST_VALU = %.4f"
                     % (100 * e["paired_fraction_of_instructions"], e["model_vs_counter"]))
        if e.get("valu_lane_instructions_per_ray_step"):
            L.append("  %.1f VALU lane-instructions per accepted ray-step; %s SIMD issue cycles per ray-step (64 rays per wave "
                     "instruction), %s of them arithmetic (fma / mul / add / transcendental)"
                     % (e["valu_lane_instructions_per_ray_step"], e.get("simd_issue_cycles_per_ray_step"),
                        e.get("simd_arithmetic_cycles_per_ray_step")))
        if e.get("flops_counted_per_ray_step"):
            L.append("  flops per ray-step as the hardware counts them: %s" % e["flops_counted_per_ray_step"])
        L.append("  %-78s %7s %6s %10s" % ("class", "share", "quads", "/ray-step"))
        for k, v in e["mix"].items():
            L.append("  %-78s %7.4f %6.1f %10s" % (k, v["share"], v["quad_cycles_each"], v["per_ray_step"]))
        L.append("  arithmetic share of the quad-cycles: %.3f" % e["arithmetic_share_of_quad_cycles"])
        L.append("")
    L += ["## calibration (tools/valu_microbench under the same counters)", "",
          "%-22s %6s %7s %7s %-12s %9s %9s %9s" % ("case", "insts", "active", "paired", "class", "cyc(w8)", "cyc(w4)", "cyc(w1)")]
    for r in cal:
        L.append("%-22s %6d %7.3f %7.3f %-12s %9s %9s %9s" % (
            r["case"], r["instructions_per_statement"], r["active_quads_per_statement"], r["paired_quads_per_statement"],
            ",".join(r["class"]) or "-", r.get("cycles_s_memtime_w8", ""), r.get("cycles_s_memtime_w4", ""),
            r.get("cycles_s_memtime_w1", "")))
    open(os.path.join(DST, "%s_valu_issue.txt" % TAG), "w").write("\n".join(L) + "\n")
    print("\n".join(L[:80]))


if __name__ == "__main__":
    main()
