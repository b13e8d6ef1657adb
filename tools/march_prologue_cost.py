#!/usr/bin/env python
"""What the part of a march launch that is NOT the march costs: the same frame with a step budget of 1
(pixel -> ray set-up, one step, shading / compositing, stores) against the full budget, for the packed WGSL
march (1080p / 512 and 8K / 1024) and the FAST GLSL march (1080p / 500).  One JSON line (GPU box)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import blackhole_simulation_amd as bh  # noqa: E402

EYE = (60.0 * np.sin(np.deg2rad(97.0)), 60.0 * np.cos(np.deg2rad(97.0)), 0.0)


def timed(fn, reps=10):
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


out = {}
with bh.PhysicsEngine(1.0, 0.999) as e:
    for name, W, H, budget in (("wgsl_packed_1080p", 1920, 1080, 512), ("wgsl_packed_8k", 7680, 4320, 1024)):
        cam = bh.camera_look_at(EYE, aspect=W / H)
        rgba = torch.zeros(W * H, 4, dtype=torch.float32, device="cuda:0")
        res = {}
        for b in (1, budget):
            wp = bh.wgsl_params(W, H, cam, 1.0, 0.999, max_steps=b, arith=bh.ARITH_FAST_PACKED)
            res["budget_%d_ms" % b] = round(timed(lambda: e.render_frame_wgsl(wp, rgba, want_total=False)), 4)
        res["share"] = round(res["budget_1_ms"] / res["budget_%d_ms" % budget], 4)
        out[name] = res
    W, H = 1920, 1080
    rgba = torch.zeros(W * H, 4, dtype=torch.float32, device="cuda:0")
    res = {}
    for b in (1, 512):
        gp = bh.glsl_params(W, H, 1.0, 0.999, max_ray_steps=b, arith=bh.ARITH_FAST)
        res["budget_%d_ms" % b] = round(timed(lambda: e.render_frame_glsl(gp, rgba, want_total=False)), 4)
    res["share"] = round(res["budget_1_ms"] / res["budget_512_ms"], 4)
    out["glsl_fast_1080p"] = res
print(json.dumps(out))
