#!/usr/bin/env python3
"""Throughput of the c3 (f64 RKF45 frame) and c2 (GLSL Verlet march) workloads across the reference's camera
range (src/configs/simulation.config.ts:106-121: angle 0.1-179.9 deg, zoom 1.5-100 R_s = 3-200 M;
src/hooks/useCamera.ts:166-167), one process, bench.py's protocol per point (c3: 3 warm-up + 20 timed frames,
one stream; c2: 60 + 300 frames, two in flight), every integrate schedule next to the others on the same camera,
with the frame's lane efficiency (sum of tries / 64 x sum over waves of the wave's slowest ray) beside it.

usage (GPU box): python tools/camera_sweep.py [--out file.jsonl] [--r0 3,10,60,200] [--theta 5,60,90,97]
                                              [--schedules one,k16] [--configs c3,c2] [--steps N]
One JSON line per (config, camera): {"config", "eye", "schedules": {name: {"G_ray_steps_per_s", "ms_per_frame"}},
"best", "default_vs_best", "lane_efficiency", "wave_max_tries": {p50, p95, p99, max}, "accepted_steps_per_frame"}"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import blackhole_simulation_amd as bh  # noqa: E402


def eye_of(r0, th_deg):
    th = np.deg2rad(th_deg)
    return (r0 * np.sin(th), r0 * np.cos(th), 0.0)


def sched_params(name):
    """schedule name -> GrvRenderParams fields"""
    if name == "one":   # the engine default: one launch, waves longest-first by the previous frame's tries
        return dict(segment_tries=0)
    if name == "slot":  # one launch in slot order (every launch before ABI 8)
        return dict(segment_tries=0, schedule=bh.SCHEDULE_SLOT_ORDER)
    if name.startswith("k"):
        return dict(segment_tries=int(name[1:]))
    raise SystemExit("unknown schedule " + name)


def time_c3(eng, W, H, eye, sched, steps, warmup, rgba):
    cam = bh.camera_look_at(eye, aspect=W / H)
    kw = sched_params(sched)
    p = bh.render_params(W, H, arith=bh.ARITH_FAST, tolerance=1e-8, **kw)
    eng.stats_accumulate(True)
    for _ in range(warmup):
        eng.render_frame_device(cam, p, rgba=rgba)
    torch.cuda.synchronize()
    eng.frame_stats_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.render_frame_device(cam, p, rgba=rgba)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng.frame_stats()
    eng.stats_accumulate(False)
    # the metric counts ACCEPTED steps; what the kernel executes is tries (rejected ones included): close to the hole the
    # controller rejects every third try (tries / steps 1.5 at r0 = 3 M, 1.02 from 60 M on), the try rate stays level
    return {"G_ray_steps_per_s": round(st.accepted_steps / dt / 1e9, 3), "ms_per_frame": round(dt / steps * 1e3, 4),
            "launches_per_frame": round(st.launches / steps, 2), "G_tries_per_s": round(st.rkf_tries / dt / 1e9, 3),
            "tries_per_accepted_step": round(st.rkf_tries / max(1, st.accepted_steps), 4),
            "disk_crossings_per_frame": int(st.crossings // steps)}, st.accepted_steps // steps


def lane_stats_c3(eng, W, H, eye):
    """per-wave statistics of the frame's integrator TRIES (what a lane executes), 8x8-pixel waves"""
    cam = bh.camera_look_at(eye, aspect=W / H)
    p = bh.render_params(W, H, arith=bh.ARITH_FAST, tolerance=1e-8, shading=0)
    steps = torch.zeros(H, W, dtype=torch.int32, device="cuda")
    eng.render_frame_device(cam, p, steps=steps)
    torch.cuda.synchronize()
    s = steps.to(torch.float64)
    hh, ww = H // 8 * 8, W // 8 * 8
    b = s[:hh, :ww].reshape(hh // 8, 8, ww // 8, 8).permute(0, 2, 1, 3).reshape(-1, 64)
    mx = b.max(dim=1).values
    return {"lane_efficiency": round(float(b.sum() / (mx.sum() * 64)), 4),
            "wave_max_steps": {k: float(mx.quantile(q)) for k, q in (("p50", 0.5), ("p95", 0.95), ("p99", 0.99), ("max", 1.0))},
            "ray_steps": {k: float(s.flatten()[::7].quantile(q)) for k, q in (("p50", 0.5), ("p95", 0.95), ("p99", 0.99))}}


def time_c2(eng, W, H, r0, th_deg, two_streams, steps, warmup, bufs):
    gp = bh.glsl_params(W, H, 1.0, 0.999, max_ray_steps=512, arith=bh.ARITH_FAST)
    gp.zoom = r0
    gp.mouse[1] = th_deg / 180.0
    streams = [torch.cuda.Stream(), torch.cuda.Stream()] if two_streams else [torch.cuda.current_stream()] * 2
    eng.stats_accumulate(True)

    def frame(i):
        with torch.cuda.stream(streams[i % 2]):
            eng.render_frame_glsl(gp, bufs[i % 2], stream=torch.cuda.current_stream().cuda_stream, want_total=False)
    for i in range(warmup):
        frame(i)
    torch.cuda.synchronize()
    eng.frame_stats_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        frame(warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng.frame_stats()
    eng.stats_accumulate(False)
    return {"G_ray_steps_per_s": round(st.accepted_steps / dt / 1e9, 3), "ms_per_frame": round(dt / steps * 1e3, 4)}, \
        st.accepted_steps // steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--r0", default="3,10,60,200")
    ap.add_argument("--theta", default="5,60,90,97")
    ap.add_argument("--schedules", default="one,k16")
    ap.add_argument("--configs", default="c3,c2")
    ap.add_argument("--steps", type=int, default=0)
    ap.add_argument("--default-schedule", default="one", help="the schedule grv_render_params_default selects")
    args = ap.parse_args()
    r0s = [float(x) for x in args.r0.split(",")]
    ths = [float(x) for x in args.theta.split(",")]
    scheds = args.schedules.split(",")
    out = open(args.out, "a") if args.out else None
    with bh.PhysicsEngine(1.0, 0.999) as eng:
        if "c3" in args.configs.split(","):
            W, H = 3840, 2160
            rgba = torch.zeros(W * H, 4, dtype=torch.float32, device="cuda")
            for r0 in r0s:
                for th in ths:
                    eye = eye_of(r0, th)
                    rec = {"config": "c3", "eye": {"r0_M": r0, "theta_deg": th}, "schedules": {}}
                    for sc in scheds:
                        rec["schedules"][sc], rec["accepted_steps_per_frame"] = time_c3(eng, W, H, eye, sc, args.steps or 20, 3, rgba)
                    rec.update(lane_stats_c3(eng, W, H, eye))
                    best = max(rec["schedules"], key=lambda k: rec["schedules"][k]["G_ray_steps_per_s"])
                    rec["best"] = best
                    d = args.default_schedule if args.default_schedule in rec["schedules"] else scheds[0]
                    rec["default_vs_best"] = round(rec["schedules"][d]["G_ray_steps_per_s"] / rec["schedules"][best]["G_ray_steps_per_s"], 4)
                    line = json.dumps(rec)
                    print(line, flush=True)
                    if out:
                        out.write(line + "\n")
                        out.flush()
        if "c2" in args.configs.split(","):
            W, H = 1920, 1080
            bufs = [torch.zeros(H, W, 4, dtype=torch.float32, device="cuda") for _ in range(2)]
            for r0 in r0s:
                for th in ths:
                    rec = {"config": "c2", "eye": {"r0_M": r0, "theta_deg": th}, "schedules": {}}
                    for name, two in (("two_streams", True), ("one_stream", False)):
                        rec["schedules"][name], rec["accepted_steps_per_frame"] = time_c2(eng, W, H, r0, th, two, args.steps or 300, 60, bufs)
                    line = json.dumps(rec)
                    print(line, flush=True)
                    if out:
                        out.write(line + "\n")
                        out.flush()


if __name__ == "__main__":
    main()
