#!/usr/bin/env python
"""Why does the FAST f32 compute march keep more rays marching to the step budget than the
shader-order march (config 4: 7680x4320, 1024 steps; profiles/r03_full_frame_parity_c4.jsonl: 4 159
vs 1 007 of 33.2 M)?  Runs on the GPU box.

The three f32 forms of the march (shader order = the oracle bit for bit, FAST, FAST packed) are
rendered with per-pixel step counts, stars off.  For every pixel where ANY form reaches the budget,
and for every pixel whose FAST colour is more than 5e-2 of peak away from the shader-order colour,
the same march is evaluated in double (oracle/wgsl_f64_twin.c: same discrete algorithm, no f32
rounding) and each f32 form is scored against it: does it agree on "still marching at 1 024", how
far is its step count, which colour is closer.
-> gpurun_out/r04_c4_budget_rays.json (committed as profiles/r04_c4_budget_rays.json)
(a checker-side measurement: it lives under tests/ because it calls the oracle)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402

import blackhole_simulation_amd as bh  # noqa: E402
import pyoracle as po  # noqa: E402

EYE = (60.0 * np.sin(np.deg2rad(97.0)), 60.0 * np.cos(np.deg2rad(97.0)), 0.0)
W, H = (int(v) for v in os.environ.get("GRV_C4_SIZE", "7680x4320").split("x"))
BUDGET = int(os.environ.get("GRV_C4_BUDGET", "1024"))
FORMS = (("shader_order", 0), ("fast", 1), ("packed", 2))


def main():
    cam = bh.camera_look_at(EYE, aspect=W / H)
    frames = {}
    with bh.PhysicsEngine(1.0, 0.999) as e:
        for name, arith in FORMS:
            gp = bh.wgsl_params(W, H, cam, 1.0, 0.999, max_steps=BUDGET, arith=arith, stars=0)
            rgba = torch.zeros(W * H, 4, dtype=torch.float32, device="cuda:0")
            steps = torch.zeros(W * H, dtype=torch.int32, device="cuda:0")
            e.render_frame_wgsl(gp, rgba, steps)
            frames[name] = (rgba.cpu().numpy().reshape(H, W, 4)[..., :3].copy(), steps.cpu().numpy().reshape(H, W).copy())
    gp0 = bh.wgsl_params(W, H, cam, 1.0, 0.999, max_steps=BUDGET, arith=0, stars=0)
    op = po.wgsl_params_from(gp0)
    threads = len(os.sched_getaffinity(0))
    peak = float(frames["shader_order"][0].max())
    out = {"frame": [W, H], "budget": BUDGET, "spin": 0.999, "peak": peak,
           "rays_at_budget": {n: int((frames[n][1] == BUDGET).sum()) for n, _ in FORMS}}

    # ---- A: pixels where any f32 form is still marching at the budget
    any_budget = np.zeros((H, W), bool)
    for n, _ in FORMS:
        any_budget |= frames[n][1] == BUDGET
    ys, xs = np.nonzero(any_budget)
    xy = np.stack([xs, ys], 1).astype(np.uint32)
    d = po.wgsl_pixels_f64(op, xy, nthreads=threads)
    f64_budget = d["cls"] == 2
    A = {"pixels": int(xy.shape[0]), "f64_at_budget": int(f64_budget.sum()),
         "f64_exit_classes": {k: int((d["cls"] == v).sum()) for k, v in (("horizon", 0), ("escape", 1), ("budget", 2), ("opaque", 3))},
         "f64_min_r_percentiles": dict(zip(("1", "50", "99"), [float(v) for v in np.percentile(d["min_r"], [1, 50, 99])])),
         # how close these rays come to the polar axis: min(theta, pi - theta) along the double march
         "f64_rays_that_step_across_the_polar_axis": int((d["axis_margin"] < 0.0).sum()),
         "f64_ended_rays_that_step_across_the_polar_axis": int(((d["axis_margin"] < 0.0) & ~f64_budget).sum()),
         "forms": {}}
    for n, _ in FORMS:
        st = frames[n][1][ys, xs].astype(np.int64)
        at = st == BUDGET
        ds = np.abs(st - d["steps"].astype(np.int64))
        A["forms"][n] = {
            "at_budget": int(at.sum()),
            "agrees_with_f64_on_budget": int((at == f64_budget).sum()),
            "at_budget_but_f64_ended": int((at & ~f64_budget).sum()),
            "ended_but_f64_at_budget": int((~at & f64_budget).sum()),
            "abs_step_difference_to_f64": {"mean": float(ds.mean()), "p50": float(np.median(ds)), "p90": float(np.percentile(ds, 90))},
            "closer_in_steps_than_shader_order": None}
    so = np.abs(frames["shader_order"][1][ys, xs].astype(np.int64) - d["steps"].astype(np.int64))
    for n in ("fast", "packed"):
        fs = np.abs(frames[n][1][ys, xs].astype(np.int64) - d["steps"].astype(np.int64))
        A["forms"][n]["closer_in_steps_than_shader_order"] = {"closer": int((fs < so).sum()), "equal": int((fs == so).sum()),
                                                              "farther": int((fs > so).sum())}
    out["A_pixels_where_any_form_reaches_the_budget"] = A

    # ---- B: pixels whose FAST colour is more than 5e-2 of peak away from the shader-order colour
    B = {}
    for n in ("fast", "packed"):
        dc = np.abs(frames[n][0] - frames["shader_order"][0]).max(-1) / peak
        ys2, xs2 = np.nonzero(dc > 5e-2)
        xy2 = np.stack([xs2, ys2], 1).astype(np.uint32)
        d2 = po.wgsl_pixels_f64(op, xy2, nthreads=threads)
        e_form = np.abs(frames[n][0][ys2, xs2] - d2["rgb"]).max(-1) / peak
        e_so = np.abs(frames["shader_order"][0][ys2, xs2] - d2["rgb"]).max(-1) / peak
        lit64 = d2["rgb"].sum(-1) > 0
        lit_f = frames[n][0][ys2, xs2].sum(-1) > 0
        lit_s = frames["shader_order"][0][ys2, xs2].sum(-1) > 0
        B[n] = {"pixels": int(xy2.shape[0]), "max_colour_difference_over_peak": float(dc.max()),
                "rays_that_step_across_the_polar_axis": int((d2["axis_margin"] < 0.0).sum()),
                "rays_within_0.05_rad_of_the_polar_axis": int((d2["axis_margin"] < 0.05).sum()),
                "rays_at_budget_in_f64": int((d2["cls"] == 2).sum()),
                "axis_margin_rad_percentiles": dict(zip(("10", "50", "90"), [float(v) for v in np.percentile(d2["axis_margin"], [10, 50, 90])])),
                "form_closer_to_f64": int((e_form < e_so).sum()), "shader_order_closer_to_f64": int((e_so < e_form).sum()),
                "tie": int((e_so == e_form).sum()),
                "lit_dark_agrees_with_f64": {"form": int((lit_f == lit64).sum()), "shader_order": int((lit_s == lit64).sum())},
                "step_count_differs_from_shader_order": int((frames[n][1][ys2, xs2] != frames["shader_order"][1][ys2, xs2]).sum()),
                "median_error_to_f64_over_peak": {"form": float(np.median(e_form)) if e_form.size else None,
                                                  "shader_order": float(np.median(e_so)) if e_so.size else None}}
    out["B_pixels_beyond_5e-2_of_peak"] = B

    # ---- C: the whole frame's step counts against the f64 twin on a 1/64 pixel subset (context)
    ys3, xs3 = np.mgrid[0:H:8, 0:W:8]
    xy3 = np.stack([xs3.ravel(), ys3.ravel()], 1).astype(np.uint32)
    d3 = po.wgsl_pixels_f64(op, xy3, nthreads=threads)
    C = {"pixels": int(xy3.shape[0]), "f64_at_budget": int((d3["cls"] == 2).sum())}
    for n, _ in FORMS:
        st = frames[n][1][::8, ::8].ravel().astype(np.int64)
        ds = np.abs(st - d3["steps"].astype(np.int64))
        C[n] = {"steps_equal_to_f64_frac": float((ds == 0).mean()), "mean_abs_step_difference": float(ds.mean()),
                "at_budget": int((st == BUDGET).sum())}
    out["C_every_8th_pixel"] = C
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r04_c4_budget_rays.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
