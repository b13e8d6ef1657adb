#!/usr/bin/env python3
"""Sweeps of what the bench lines hold fixed (run on the GPU box; the camera sweep is tools/camera_sweep.py):

  --resolution    the f64 frame (BASELINE configs[2] at other sizes) and both f32 marches from 8K / 4K down to 360p,
                  one frame at a time and two in flight               -> profiles/r06_ab_small_frame_order.jsonl (its second half)
  --tolerance     the 4K f64 frame at RKF45 tolerances 1e-5 ... 1e-10
  --glsl          the 1080p GLSL march over step budgets, feature bits, disk sizes / heights, turbulence, overlays, cameras

One JSON line per case on stdout.  Round 6 found three things this way (profiles/EXPERIMENTS.md W, X, Y): a sweep is
cheap, and every size or camera a decision was tuned on is the only one it is known to be right for."""
import argparse
import json
import os
import subprocess
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench(*args):
    out = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--no-cpu-baseline", *args], capture_output=True, text=True, timeout=600)
    d = json.loads(out.stdout.strip().splitlines()[-1])
    return {"args": " ".join(args), "G_ray_steps_per_s": round(d["value"] / 1e3, 2), "ms_per_frame": d["ms_per_step"],
            "kernel_ms": d["roofline"].get("avg_launch_ms"), "accepted_steps_per_frame": d["config"]["accepted_steps_per_frame"]}


def resolution():
    for w, h in ((7680, 4320), (3840, 2160), (2560, 1440), (1920, 1080), (1280, 720), (960, 540), (640, 360)):
        for extra in ((), ("--two-streams",)):
            print(json.dumps(bench("--width", str(w), "--height", str(h), "--steps", "40", "--warmup", "5", *extra)), flush=True)
    for w, h in ((3840, 2160), (2560, 1440), (1920, 1080), (1280, 720), (960, 540), (640, 360)):
        for extra in ((), ("--one-stream",), ("--kernel", "wgsl", "--one-stream")):
            print(json.dumps(bench("--config", "c2", "--width", str(w), "--height", str(h), "--steps", "200", "--warmup", "40", *extra)), flush=True)


def tolerance():
    for tol in ("1e-5", "1e-6", "1e-7", "1e-8", "1e-9", "1e-10"):
        print(json.dumps(bench("--config", "c3", "--tolerance", tol, "--steps", "10", "--warmup", "2")), flush=True)


def glsl():
    sys.path.insert(0, R)
    import torch
    import blackhole_simulation_amd as bh
    W, H = 1920, 1080

    def run(e, name, **kw):
        theta = kw.pop("theta", None)
        gp = bh.glsl_params(W, H, 1.0, 0.999, arith=bh.ARITH_FAST, **kw)
        if theta is not None:
            gp.mouse[1] = theta / 180.0
        buf = torch.zeros(H, W, 4, dtype=torch.float32, device="cuda")
        e.stats_accumulate(True)
        for _ in range(30):
            e.render_frame_glsl(gp, buf, want_total=False)
        torch.cuda.synchronize()
        e.frame_stats_reset()
        torch.cuda.synchronize()
        t, n = time.perf_counter(), 150
        for _ in range(n):
            e.render_frame_glsl(gp, buf, want_total=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        st = e.frame_stats()
        e.stats_accumulate(False)
        print(json.dumps({"case": name, "ms_per_frame": round(dt / n * 1e3, 4), "G_ray_steps_per_s": round(st.accepted_steps / dt / 1e9, 1),
                          "steps_per_pixel": round(st.accepted_steps / n / (W * H), 1)}), flush=True)

    with bh.PhysicsEngine(1.0, 0.999) as e:
        D = bh.GLSL_FEATURES_DEFAULT
        for ms in (512, 256, 128, 64):
            run(e, "default preset, %d steps" % ms, max_ray_steps=ms)
        for name, bit in (("lensing", 1), ("disk", 2), ("doppler", 4), ("stars", 8), ("glow", 16), ("jets", 32), ("redshift", 64), ("dither", 128)):
            run(e, "default %s %s" % ("without" if D & bit else "with", name), max_ray_steps=512, features=D ^ bit)
        run(e, "all features", max_ray_steps=512, features=255)
        for ds in (6.0, 15.0, 40.0):
            run(e, "disk_size %g" % ds, max_ray_steps=512, disk_size=ds)
        for dh in (0.05, 0.2, 0.45):
            run(e, "disk_scale_height %g" % dh, max_ray_steps=512, disk_scale_height=dh)
        run(e, "constant turbulence 0.75 (no noise)", max_ray_steps=512, turbulence=0.75)
        run(e, "time 137", max_ray_steps=512, time=137.0)
        run(e, "show_redshift", max_ray_steps=512, show_redshift=1.0)
        run(e, "show_kerr_shadow", max_ray_steps=512, show_kerr_shadow=1.0)
        run(e, "lensing_strength 0.5", max_ray_steps=512, lensing_strength=0.5)
        for th in (5.0, 60.0, 80.0, 90.0):
            run(e, "theta %g" % th, max_ray_steps=512, theta=th)
        for z in (3.0, 10.0, 30.0, 200.0):
            run(e, "zoom %g" % z, max_ray_steps=512, zoom=z)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--resolution", action="store_true")
    ap.add_argument("--tolerance", action="store_true")
    ap.add_argument("--glsl", action="store_true")
    a = ap.parse_args()
    if not (a.resolution or a.tolerance or a.glsl):
        ap.error("choose at least one sweep")
    if a.resolution:
        resolution()
    if a.tolerance:
        tolerance()
    if a.glsl:
        glsl()
