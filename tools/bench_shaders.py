"""Secondary measurements (not the bench.py line): the f32 shader kernels on BASELINE
configs[1] (1920x1080, 512 max steps) and configs[3]'s per-GPU share (7680x4320 / 8 ranks,
1024 steps; the GLSL shader itself caps at 500).  Run on the GPU box:
    python tools/bench_shaders.py
Prints one JSON line per case: ms/frame (HIP-event-free wall time over synchronised frames),
Mray-steps/s, algorithmic GB/s at 72 B per f32 ray-step."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import blackhole_simulation_amd as bh  # noqa: E402

EYE = (60.0 * np.sin(np.deg2rad(97.0)), 60.0 * np.cos(np.deg2rad(97.0)), 0.0)


def run(name, W, H, make, call, reps=10, world=1):
    gp = make(W, H)
    gp.tile_world, gp.tile_rank = world, 0
    n = bh.load_library().grv_frame_ray_count(bh.render_params(W, H, tile_world=world, tile_rank=0))
    rgba = torch.zeros(n, 4, dtype=torch.float32, device="cuda:0")
    with bh.PhysicsEngine(1.0, 0.999) as e:
        tot = call(e, gp, rgba)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            tot = call(e, gp, rgba)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / reps * 1e3
    print(json.dumps({"case": name, "width": W, "height": H, "tile_world": world, "rays": n,
                      "steps_per_frame": int(tot), "ms_per_frame": round(ms, 3),
                      "Mray_steps_per_s": round(tot / ms / 1e3, 1),
                      "algorithmic_GBps_72B": round(tot * 72 / ms / 1e6, 1)}), flush=True)


def wgsl(max_steps, arith=0):
    def make(W, H):
        cam = bh.camera_look_at(EYE, aspect=W / H)
        return bh.wgsl_params(W, H, cam, 1.0, 0.999, max_steps=max_steps, arith=arith)
    return make


def glsl(max_steps, **kw):
    return lambda W, H: bh.glsl_params(W, H, 1.0, 0.999, max_ray_steps=max_steps, **kw)


if __name__ == "__main__":
    cw = lambda e, gp, rgba: e.render_frame_wgsl(gp, rgba)  # noqa: E731
    cg = lambda e, gp, rgba: e.render_frame_glsl(gp, rgba)  # noqa: E731
    run("C2 wgsl symplectic f32, 512 steps", 1920, 1080, wgsl(512), cw)
    run("C2 wgsl symplectic f32 FAST, 512 steps", 1920, 1080, wgsl(512, 1), cw)
    run("C2 wgsl symplectic f32 FAST packed (2 rays/lane), 512 steps", 1920, 1080, wgsl(512, 2), cw)
    run("C2 glsl verlet f32 (march+disk), 512->500 steps", 1920, 1080, glsl(512, features=7, turbulence=0.75), cg)
    run("C2 glsl full default preset, 512->500 steps", 1920, 1080, glsl(512), cg)
    run("C2 glsl verlet f32 FAST (march+disk), 512->500 steps", 1920, 1080,
        glsl(512, features=7, turbulence=0.75, arith=1), cg)
    run("C2 glsl full default preset FAST, 512->500 steps", 1920, 1080, glsl(512, arith=1), cg)
    run("C4 share wgsl, 8K / 8 ranks, 1024 steps", 7680, 4320, wgsl(1024), cw, reps=5, world=8)
    run("C4 share wgsl FAST, 8K / 8 ranks, 1024 steps", 7680, 4320, wgsl(1024, 1), cw, reps=5, world=8)
    run("C4 share glsl full, 8K / 8 ranks, 1024->500 steps", 7680, 4320, glsl(1024), cg, reps=5, world=8)
    run("C4 share glsl full FAST, 8K / 8 ranks, 1024->500 steps", 7680, 4320, glsl(1024, arith=1), cg,
        reps=5, world=8)
    run("4K wgsl, 150 steps (shader default)", 3840, 2160, wgsl(150), cw)
    run("4K wgsl FAST, 150 steps (shader default)", 3840, 2160, wgsl(150, 1), cw)
    run("4K wgsl FAST packed (2 rays/lane), 150 steps", 3840, 2160, wgsl(150, 2), cw)
