"""Turns the rocprofv3 rocpd databases under gpurun_out/prof into the committed
text/JSON summaries under profiles/ (kernel-trace stats + PMC counters)."""
import json
import os
import sqlite3
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
DST = sys.argv[2] if len(sys.argv) > 2 else "profiles"
TAG = sys.argv[3] if len(sys.argv) > 3 else "r01"
os.makedirs(DST, exist_ok=True)


def q(db, sql):
    return sqlite3.connect(db).execute(sql).fetchall()


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    depth = 0
    for k, ch in enumerate(n):  # cut the argument list, keep template arguments
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return n[:k]
    return n


def trace_summary(dbpath, label):
    rows = q(dbpath, "select name, total_calls, total_duration, average, percentage from top_kernels")
    lines = ["# rocprofv3 --kernel-trace --stats  (%s)" % label,
             "%-60s %8s %14s %14s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for n, c, t, a, p in rows:
        lines.append("%-60s %8d %14.1f %14.2f %7.2f" % (short(n)[:60], c, t, a, p))
    reg = q(dbpath, "select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, "
                    "workgroup_x, min(grid_x), max(grid_x), count(*), avg(duration), min(duration), max(duration) "
                    "from kernels group by name")
    lines.append("")
    lines.append("%-44s %5s %5s %5s %7s %7s %5s %10s %10s %6s %12s %12s %12s" % (
        "kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "wg", "grid_min", "grid_max", "n", "avg_ns", "min_ns", "max_ns"))
    for r in reg:
        lines.append("%-44s %5d %5d %5d %7d %7d %5d %10d %10d %6d %12.0f %12d %12d" % ((short(r[0])[:44],) + tuple(r[1:])))
    return "\n".join(lines) + "\n", rows


def pmc_summary(dbpath):
    rows = q(dbpath, "select kernel_name, counter_name, count(*), avg(value), sum(value) from "
                     "counters_collection group by kernel_name, counter_name")
    return rows


out = {}
TRACE_CMD = {"trace": "--steps 10 --warmup 2", "trace_k16": "--steps 10 --warmup 2 --segment-tries 16",
             "trace_strict": "--steps 5 --warmup 1 --arith strict", "trace_c4": "--steps 5 --warmup 1 --config c4",
             "trace_c4fast": "--steps 5 --warmup 1 --config c4 --arith fast",
             "trace_c2": "--steps 5 --warmup 1 --config c2", "trace_c2wgsl": "--steps 5 --warmup 1 --config c2 --kernel wgsl",
             "trace_c5": "--steps 5 --warmup 1 --config c5"}
for label in TRACE_CMD:
    p = os.path.join(SRC, label, "bench_results.db")
    if os.path.exists(p):
        txt, rows = trace_summary(p, "python bench.py %s --no-cpu-baseline" % TRACE_CMD[label])
        bj = os.path.join(SRC, label + "_bench.json")
        if os.path.exists(bj):
            txt += "\n# bench.py line of the same run\n" + open(bj).read().strip() + "\n"
        open(os.path.join(DST, "%s_%s_kernel_stats.txt" % (TAG, label)), "w").write(txt)
        print(txt)

pm = {}
PMC_LABELS = ["pmc_fetch_k16", "pmc_write_k16"] + [
    "pmc_%s%s" % (kind, sfx) for sfx in ("", "_strict", "_c4", "_c4fast", "_c2", "_c2wgsl")
    for kind in ("fetch", "write", "sq", "cls32", "cls64", "occ", "mem")]
for label in PMC_LABELS:
    p = os.path.join(SRC, label, "bench_results.db")
    if os.path.exists(p):
        for k, c, n, avg, tot in pmc_summary(p):
            pm.setdefault(label, []).append(dict(kernel=short(k), counter=c, dispatches=n, avg=avg, total=tot))
open(os.path.join(DST, "%s_pmc_counters.json" % TAG), "w").write(json.dumps(pm, indent=1))
for label, items in pm.items():
    for it in items:
        if any(w in it["kernel"] for w in ("integrate", "finalize", "init_from", "wgsl", "glsl")):
            print(label, it)

# HBM traffic of the dominant kernels per launch (bench.py reads profiles/traffic.json).  Every
# entry carries the code hash of the kernel it was measured on (code_hashes.json is written on the
# GPU box by tools/profile_gpu.sh from the library that actually ran): bench.py quotes an entry
# only for a library whose kernel hashes the same.
def _avg(label, counter, needle):
    for it in pm.get(label, []):
        if needle in it["kernel"] and it["counter"] == counter:
            return it["avg"]
    return None


def _bench_line(label):
    """the bench.py line a pass printed (one JSON line), or None"""
    try:
        for ln in open(os.path.join(SRC, label + "_bench.json")):
            if ln.startswith("{"):
                return json.loads(ln)
    except Exception:
        pass
    return None


def _steps_per_launch(label):
    ln = _bench_line(label)
    if not ln:
        return None
    try:
        return int(ln["config"]["accepted_steps_per_frame"] / max(ln["roofline"]["launches_per_frame"], 1.0))
    except Exception:
        return None


# every entry names the pass it came from: "<tag>-<session directory>"; bench.py prints it
# (roofline.traffic_pass), so a committed bench record says which counter pass it quoted
PASS_ID = "%s-%s" % (TAG, os.path.basename(os.path.normpath(SRC)))
hashes = {}
hp = os.path.join(SRC, "code_hashes.json")
if os.path.exists(hp):
    hashes = json.load(open(hp))
tpath = os.path.join(DST, "traffic.json")
traffic = {"format": 3, "kernels": {}}
if os.path.exists(tpath):
    old = json.load(open(tpath))
    if old.get("format") in (2, 3):
        traffic = old
        traffic["format"] = 3

# (pretty name, substring of the demangled name in the trace, pass suffix, frame, layout bytes)
CASES = [("integrate_segment_kernel<1,1,0>", "integrate_segment_kernel<1, 1, 0>", "", [3840, 2160], 8355840 * (92 + 76)),
         ("integrate_segment_kernel<1,0,0>", "integrate_segment_kernel<1, 0, 0>", "_strict", [3840, 2160], 8355840 * (92 + 76)),
         ("wgsl_symplectic_pk_kernel", "wgsl_symplectic_pk_kernel", "_c4", [7680, 4320], None),
         ("wgsl_symplectic_fast_kernel", "wgsl_symplectic_fast_kernel", "_c4fast", [7680, 4320], None),
         # BASELINE configs[1]: the 1080p / 512-step marches (a kernel measured at a second frame size is filed
         # under "<kernel>@<W>x<H>"; bench.py looks that key up first)
         ("glsl_fragment_kernel<1>", "glsl_fragment_kernel<1>", "_c2", [1920, 1080], None),
         ("wgsl_symplectic_pk_kernel@1920x1080", "wgsl_symplectic_pk_kernel", "_c2wgsl", [1920, 1080], None)]
for pretty, needle, sfx, frame, layout in CASES:
    f, w = _avg("pmc_fetch" + sfx, "FETCH_SIZE", needle), _avg("pmc_write" + sfx, "WRITE_SIZE", needle)
    if f is None or w is None:
        continue
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  gfx950 correction
    # (/opt/skills/guides/MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request
    # on coalesced streaming reads -> x2.  Cross-check for the f64 frame: 8 355 840 slots x 92 B
    # read = 0.769 GB, x 76 B written = 0.635 GB.
    ent = {"name": needle, "code_hash": hashes.get(pretty.split("@")[0]), "frame": frame, "pass_id": PASS_ID,
           # accepted ray-steps one launch of the profiled workload processed (bench line of the pmc_occ run)
           "ray_steps_per_launch": _steps_per_launch("pmc_occ" + sfx) or _steps_per_launch("pmc_fetch" + sfx),
           "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline" +
                      {"": "", "_strict": " --arith strict", "_c4": " --config c4", "_c4fast": " --config c4 --arith fast",
                       "_c2": " --config c2", "_c2wgsl": " --config c2 --kernel wgsl"}[sfx],
           "fetch_size_kib_raw_avg_per_launch": f, "write_size_kib_avg_per_launch": w,
           "fetch_correction": 2.0, "hbm_bytes_per_launch": int((2.0 * f + w) * 1024)}
    if layout:
        ent["expected_from_layout_bytes"] = layout
    if sfx == "":
        # (since round 6 every launch of the compacting schedule is the compact kernel: the live count comes from device memory)
        k16 = needle.replace("integrate_segment_kernel", "integrate_compact_kernel")
        f2, w2 = _avg("pmc_fetch_k16", "FETCH_SIZE", k16), _avg("pmc_write_k16", "WRITE_SIZE", k16)
        if f2 is not None and w2 is not None:
            ent["segment_tries_16"] = {"fetch_size_kib_raw_avg_per_launch": f2,
                                       "write_size_kib_avg_per_launch": w2,
                                       "hbm_bytes_per_launch": int((2.0 * f2 + w2) * 1024),
                                       "launches_per_frame": ((_bench_line("pmc_fetch_k16") or {}).get("roofline") or {})
                                       .get("launches_per_frame", 32),
                                       "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE -- python bench.py "
                                                  "--steps 2 --warmup 1 --no-cpu-baseline --segment-tries 16"}
    # what actually bounds the kernel: VALU issue.  A gfx950 SIMD issues one VALU instruction per
    # quad-cycle (4 cycles) -- f64, packed f32, conversions, compares, selects alike --, transcendentals
    # hold it 2 (f32) / 4 (f64) quad-cycles, and two full-rate 32-bit ops can share one (calibrated with
    # tools/valu_microbench, profiles/*_valu_issue.txt).  SQ_ACTIVE_INST_VALU counts the quad-cycles
    # instruction by instruction, SQ_ACTIVE_INST_VALU2 the shared ones, so
    #   occupancy = (ACTIVE_INST_VALU - ACTIVE_INST_VALU2) * 4 / (1024 SIMDs * elapsed cycles),
    # elapsed = GRBM_GUI_ACTIVE / 8 XCDs, all three from ONE run (pmc_occ).  Flops as the hardware counts
    # them: SQ_INSTS_VALU_FLOPS_* is per wave instruction (fma 2, packed fma 4, mul / add / trans 1) -> x64 lanes.
    occ = {c: _avg("pmc_occ" + sfx, c, needle) for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VALU2",
                                                        "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU_FLOPS_FP32",
                                                        "SQ_INSTS_VALU_FLOPS_FP64")}
    if occ["SQ_ACTIVE_INST_VALU"] and occ["GRBM_GUI_ACTIVE"] and occ["SQ_ACTIVE_INST_VALU2"] is not None:
        cyc = occ["GRBM_GUI_ACTIVE"] / 8.0
        flops = 64.0 * ((occ["SQ_INSTS_VALU_FLOPS_FP32"] or 0.0) + (occ["SQ_INSTS_VALU_FLOPS_FP64"] or 0.0))
        ent["valu"] = {"sq_insts_valu_per_launch": occ["SQ_INSTS_VALU"],
                       "sq_active_inst_valu": occ["SQ_ACTIVE_INST_VALU"], "sq_active_inst_valu2": occ["SQ_ACTIVE_INST_VALU2"],
                       "cycles_per_xcd": cyc,
                       "issue_frac": round((occ["SQ_ACTIVE_INST_VALU"] - occ["SQ_ACTIVE_INST_VALU2"]) * 4.0 / 1024.0 / cyc, 4),
                       "flops_counted_per_launch": flops,
                       "note": "(SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2) x 4 cycles / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8); "
                               "flops = SQ_INSTS_VALU_FLOPS_FP32/64 x 64 lanes"}
    else:
        insts, gui = _avg("pmc_sq" + sfx, "SQ_INSTS_VALU", needle), _avg("pmc_sq" + sfx, "GRBM_GUI_ACTIVE", needle)
        act = _avg("pmc_sq" + sfx, "SQ_ACTIVE_INST_VALU", needle)
        act2 = _avg("pmc_cls64" + sfx, "SQ_ACTIVE_INST_VALU2", needle)
        if insts and gui and act and act2 is not None:
            ent["valu"] = {"sq_insts_valu_per_launch": insts, "cycles_per_xcd": gui / 8.0,
                           "issue_frac": round((act - act2) * 4.0 / 1024.0 / (gui / 8.0), 4),
                           "note": "(SQ_ACTIVE_INST_VALU - SQ_ACTIVE_INST_VALU2) x 4 cycles / 1024 SIMDs / elapsed cycles "
                                   "(the two counters from two runs)"}
    mem = {c: _avg("pmc_mem" + sfx, c, needle) for c in ("SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_FLAT",
                                                        "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM",
                                                        "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")}
    if any(v is not None for v in mem.values()):
        ent["memory_side_instructions_per_launch"] = mem
    traffic["kernels"][pretty] = ent
open(tpath, "w").write(json.dumps(traffic, indent=1))
print(json.dumps(traffic, indent=1))
