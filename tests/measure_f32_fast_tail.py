#!/usr/bin/env python
"""Where do the f32 FAST marches differ from the shader-order oracle, and by how much?
Runs on the GPU box.  For the WGSL compute march (FAST and two-rays-per-lane PACKED) at the test
size and at BASELINE configs[3] (7680x4320, 1024 steps, every 16th pixel in x and y), stars off:
percentiles of |d steps| and |d colour| / peak, the fractions beyond the two colour bars, and a
classification of the pixels beyond 5e-2 (step count differs / lit-dark flip at a disk edge / other).
-> profiles/r03_f32_fast_tail.json
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


def stats(got_rgba, got_steps, ref_rgba, ref_steps):
    ds = np.abs(got_steps.astype(np.int64) - ref_steps.astype(np.int64)).reshape(-1)
    peak = max(float(ref_rgba[..., :3].max()), 1e-12)
    dc = (np.abs(got_rgba - ref_rgba)[..., :3].max(-1) / peak).reshape(-1)
    q = [50, 90, 99, 99.9, 99.99, 100]
    lit_g = got_rgba[..., :3].sum(-1).reshape(-1) > 0
    lit_r = ref_rgba[..., :3].sum(-1).reshape(-1) > 0
    bad = dc > 5e-2
    flip = lit_g != lit_r
    return {
        "pixels": int(dc.size), "peak": peak,
        "steps_equal_frac": float((ds == 0).mean()), "steps_within_2_frac": float((ds <= 2).mean()),
        "dsteps_percentiles": dict(zip(map(str, q), [float(np.percentile(ds, x)) for x in q])),
        "dcolour_percentiles": dict(zip(map(str, q), [float(np.percentile(dc, x)) for x in q])),
        "frac_beyond_2e-3": float((dc > 2e-3).mean()), "frac_beyond_5e-2": float(bad.mean()),
        "beyond_5e-2": {"count": int(bad.sum()), "with_step_count_differing": int((bad & (ds != 0)).sum()),
                        "lit_dark_flip": int((bad & flip).sum()),
                        "same_steps_same_litness": int((bad & (ds == 0) & ~flip).sum())},
        "identical_pixel_frac": float((got_rgba == ref_rgba).all(-1).mean()),
    }


def run(W, H, spin, max_steps, arith, stride):
    cam = bh.camera_look_at(EYE, aspect=W / H)
    gp = bh.wgsl_params(W, H, cam, 1.0, spin, max_steps=max_steps, arith=arith, stars=0)
    if stride == 1:
        gp.jitter[0], gp.jitter[1] = 0.0, -1.0 / 6.0
    with bh.PhysicsEngine(1.0, spin) as e:
        rgba = torch.zeros(W * H, 4, dtype=torch.float32, device="cuda:0")
        steps = torch.zeros(W * H, dtype=torch.int32, device="cuda:0")
        e.render_frame_wgsl(gp, rgba, steps)
    g = rgba.cpu().numpy().reshape(H, W, 4)[::stride, ::stride]
    s = steps.cpu().numpy().reshape(H, W)[::stride, ::stride]
    r, rs = po.wgsl_frame(po.wgsl_params_from(gp), stride=(stride, stride), nthreads=os.cpu_count() or 8)
    return stats(g, s, r, rs)


def main():
    out = {"what": __doc__.split("\n")[0], "cases": []}
    for (W, H, spin, ms, stride) in [(480, 270, 0.999, 512, 1), (480, 270, 0.5, 150, 1), (1920, 1080, 0.999, 512, 4),
                                     (7680, 4320, 0.999, 1024, 16)]:
        for arith, name in ((1, "fast"), (2, "packed")):
            st = run(W, H, spin, ms, arith, stride)
            st.update(frame=[W, H], spin=spin, max_steps=ms, arith=name, stride=stride)
            out["cases"].append(st)
            print(json.dumps(st), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r03_f32_fast_tail.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
