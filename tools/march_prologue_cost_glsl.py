#!/usr/bin/env python
"""Where the non-march part of the FAST GLSL frame goes: budget-1 frames (pixel -> ray set-up, one step, background,
glow, stores) with features switched off one by one, and the full-budget frame beside them.  One JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import blackhole_simulation_amd as bh  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


W, H = 1920, 1080
out = {}
D = bh.GLSL_FEATURES_DEFAULT
cases = {"default": D, "no_stars": D & ~bh.GLSL_STARS, "no_glow": D & ~bh.GLSL_PHOTON_GLOW, "no_dither": D & ~bh.GLSL_DITHER,
         "no_jets": D & ~bh.GLSL_JETS, "no_disk": D & ~(bh.GLSL_DISK | bh.GLSL_JETS), "lensing_only": bh.GLSL_LENSING}
with bh.PhysicsEngine(1.0, 0.999) as e:
    rgba = torch.zeros(W * H, 4, dtype=torch.float32, device="cuda:0")
    for budget in (1, 512):
        for name, feat in cases.items():
            gp = bh.glsl_params(W, H, 1.0, 0.999, max_ray_steps=budget, arith=bh.ARITH_FAST, features=feat)
            out["%s_budget_%d_ms" % (name, budget)] = round(timed(lambda: e.render_frame_glsl(gp, rgba, want_total=False)), 4)
print(json.dumps(out))
