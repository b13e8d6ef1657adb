#!/usr/bin/env python
"""Wave-by-wave timeline of one launch of the FAST GLSL march (config c2's frame), from an A/B library
built with -DGRV_MARCH_TIMELINE (tools/ab_build.sh timeline "-DGRV_MARCH_TIMELINE"; the product library
has no such hook).  Prints one JSON record: launch length, waves in flight over time (ramp / plateau /
tail), how long the launch runs with less than 1/2, 1/4 of the chip's wave slots busy, and the spread
of the per-XCC finish times.

    python tools/march_timeline.py ab_libs/lib_timeline.so [--frames 3] [--width 1920 --height 1080]
"""
import argparse
import ctypes as C
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    args = ap.parse_args()
    target = os.path.join(ROOT, "blackhole-simulation_amd", "libgravitas_hip.so")
    keep = target + ".orig"
    shutil.copy(target, keep)
    shutil.copy(args.lib, target)
    try:
        sys.path.insert(0, ROOT)
        import torch
        import blackhole_simulation_amd as bh
        L = bh.load_library()
        W, H = args.width, args.height
        n_waves = ((W + 63) // 64) * ((H + 63) // 64) * 64
        buf = torch.zeros(n_waves, 4, dtype=torch.int64, device="cuda:0")
        L.grv_debug_set_march_timeline.argtypes = [C.c_void_p]
        assert L.grv_debug_set_march_timeline(C.c_void_p(buf.data_ptr())) == 0
        out = torch.zeros(W * H, 4, dtype=torch.float32, device="cuda:0")
        recs = []
        with bh.PhysicsEngine(1.0, 0.999) as e:
            gp = bh.glsl_params(W, H, 1.0, 0.999, max_ray_steps=512, arith=bh.ARITH_FAST)
            for f in range(args.frames + 1):
                buf.zero_()
                torch.cuda.synchronize()
                e.render_frame_glsl(gp, out, want_total=False)
                torch.cuda.synchronize()
                if f == 0:
                    continue  # warm-up
                t = buf.cpu().numpy()
                t = t[t[:, 1] > 0]
                t0, t1 = t[:, 0].astype(np.float64), t[:, 1].astype(np.float64)
                tick = 1e-8  # s_memrealtime: 100 MHz
                start, end = t0.min(), t1.max()
                length = (end - start) * tick
                grid = np.linspace(start, end, 401)
                active = np.array([((t0 <= g) & (t1 > g)).sum() for g in grid])
                peak = active.max()
                dt = length / 400
                xcc = (t[:, 2] >> 32) & 0xF
                fin = [float((t1[xcc == x].max() - start) * tick * 1e3) for x in sorted(set(xcc.tolist()))]
                recs.append({
                    "launch_ms": round(length * 1e3, 4), "waves": int(t.shape[0]), "peak_waves_in_flight": int(peak),
                    "ms_below_half_peak_at_start": round(float((active[:200] < peak / 2).sum() * dt * 1e3), 4),
                    "ms_below_half_peak_at_end": round(float((active[200:] < peak / 2).sum() * dt * 1e3), 4),
                    "ms_below_quarter_peak_at_end": round(float((active[200:] < peak / 4).sum() * dt * 1e3), 4),
                    "mean_waves_in_flight_over_peak": round(float(active.mean() / peak), 4),
                    "wave_ms": {"p50": round(float(np.median(t1 - t0) * tick * 1e3), 4),
                                "p99": round(float(np.percentile(t1 - t0, 99) * tick * 1e3), 4),
                                "max": round(float((t1 - t0).max() * tick * 1e3), 4)},
                    "last_wave_started_ms": round(float((t0.max() - start) * tick * 1e3), 4),
                    "xcc_finish_ms": [round(x, 4) for x in fin],
                    "steps_of_the_last_32_waves_to_finish": [int(x) for x in t[np.argsort(t1)[-32:], 3] // 64],
                    "in_flight_profile_20": [int(x) for x in active[::20]],
                })
        print(json.dumps({"frame": "%dx%d c2 default preset, FAST GLSL march" % (W, H), "launches": recs}))
    finally:
        shutil.copy(keep, target)
        os.remove(keep)
        sys.stdout.flush()
        os._exit(0)  # the instrumented library is unmapped under a live HIP runtime otherwise


if __name__ == "__main__":
    main()
