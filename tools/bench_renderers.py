"""Secondary measurement: whole frames of the two renderers (scene + post chain) per second,
against the reference's design target of 60-144 FPS (docs/PERFORMANCE.md:3).
Run on the GPU box: python tools/bench_renderers.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

import blackhole_simulation_amd as bh  # noqa: E402


def blocks(W, H, eye):
    c = bh.camera_look_at(eye, aspect=W / H)
    iv = np.array(c.inv_view, np.float64).reshape(4, 4).T
    ip = np.array(c.inv_proj, np.float64).reshape(4, 4).T
    view, proj = np.linalg.inv(iv), np.linalg.inv(ip)
    cu = np.zeros(88, np.float32)
    for k, m in enumerate((view, proj, iv, ip, proj @ view)):
        cu[16 * k:16 * k + 16] = m.T.reshape(-1)
    cu[80:83] = eye
    pp = np.zeros(8, np.float32)
    pp[0], pp[1], pp[2], pp[3] = 1.0, 0.9, W, H
    return cu, pp


def fps(fn, reps):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return reps / (time.perf_counter() - t)


if __name__ == "__main__":
    eye = (59.55, -7.31, 0.0)
    with bh.PhysicsEngine(1.0, 0.9) as e:
        for W, H in ((1920, 1080), (3840, 2160)):
            screen = torch.zeros(H, W, 4, dtype=torch.float32, device="cuda:0")
            cu, pp = blocks(W, H, eye)
            for arith, name in ((bh.ARITH_STRICT, "shader order"), (bh.ARITH_FAST, "fast"),
                                (2, "fast, WebGPU march with two rays per lane")):
                e.renderer_reset()
                f1 = fps(lambda: e.webgpu_render(cu, pp, screen, max_steps=150, arith=arith), 20)
                gp = bh.glsl_params(W, H, 1.0, 0.9, max_ray_steps=256, arith=min(arith, 1))   # "ultra" budget
                e.renderer_reset()
                f2 = fps(lambda: e.webgl_render(gp, screen, bloom=True), 20)
                print(json.dumps({"width": W, "height": H, "arith": name,
                                  "webgpu_render_fps (150 steps, ATAA, blit)": round(f1, 1),
                                  "webgl_render_fps (256 steps, default preset, TAA, bloom)": round(f2, 1)}),
                      flush=True)
