"""Generates tests/golden/shaders_v1.npz from the CPU oracle: small frames of the two f32
shader restatements (WGSL compute march, GLSL fragment shader with each compositing feature),
the post chain on a seeded image, and the spacetime read-outs.  The reference cannot run these
shaders here (WebGL/WebGPU; its own tests mock the GL context), so the
vectors pin the restatements against accidental change (CPU test) and are compared with the
HIP kernels on the GPU box (tests/test_golden_shaders.py).

    python tests/golden/make_golden_shaders.py      # rewrites tests/golden/shaders_v1.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402

import blackhole_simulation_amd as bh  # noqa: E402  (host-side parameter helpers only; no GPU)

W, H = 64, 36
EYE = (60.0 * np.sin(np.deg2rad(97.0)), 60.0 * np.cos(np.deg2rad(97.0)), 0.0)

# name -> GrvGlslParams overrides (spin, kwargs)
GLSL_CASES = {
    "march_disk": (0.999, dict(features=7, turbulence=0.75)),
    "default_preset": (0.9, dict()),
    "animated": (0.9, dict(time=3.7, tone_map=1)),
    "redshift_overlay": (-0.7, dict(features=7 | 64, show_redshift=1.0)),
    "low_quality": (0.9, dict(quality=0)),
    "sab_camera": (0.9, dict(cam_pos=(0.0, 6.0, -60.0), cam_quat=(0.05, 0.0, 0.0, 0.99875), tone_map=1)),
}


def glsl_case(name):
    spin, kw = GLSL_CASES[name]
    return bh.glsl_params(W, H, 1.0, spin, max_ray_steps=300, **kw)


def post_image(seed=7, h=24, w=40, hdr=4.0):
    rng = np.random.default_rng(seed)
    img = rng.random((h, w, 4), dtype=np.float32) ** 3 * np.float32(hdr)
    img[..., 3] = 1.0
    return img


def main():
    out = {}
    cam = bh.camera_look_at(EYE, aspect=W / H)
    gp = bh.wgsl_params(W, H, cam, 1.0, 0.999, max_steps=300, stars=0)
    gp.jitter[0], gp.jitter[1] = 0.0, -1.0 / 6.0
    rgba, steps = po.wgsl_frame(po.wgsl_params_from(gp), nthreads=4)
    out["wgsl_rgba"], out["wgsl_steps"] = rgba, steps
    for name in GLSL_CASES:
        rgba, steps = po.glsl_frame(po.glsl_params_from(glsl_case(name)), nthreads=4)
        out["glsl_%s_rgba" % name], out["glsl_%s_steps" % name] = rgba, steps
    cur, hist = post_image(7), post_image(8)
    out["post_taa"] = po.taa_resolve(cur, hist, 0.75, False, True)
    out["post_taa_moving"] = po.taa_resolve(cur, hist, 0.75, True, True)
    out["post_bloom"] = po.bloom(post_image(9, hdr=6.0), 0.8, 0.5, 2, True)
    out["viz_curvature"] = po.scalar_field(0, 1.0, 0.9, 2.2, 40.0, 9, 7)
    out["viz_tilt"] = po.scalar_field(1, 1.0, 0.9, 2.2, 40.0, 9, 7)
    out["viz_frame_drag"] = po.scalar_field(2, 1.0, 0.9, 2.2, 40.0, 9, 7)
    out["viz_embedding"] = po.embedding_mesh(1.0, 0.9, 2.5, 30.0, 8, 6)
    out["viz_ergosphere"] = po.ergosphere_mesh(1.0, 0.9, 9, 6)
    out["noise_probe"] = po.seeded_noise_rgba8(1)[:64].copy()
    np.savez_compressed(os.path.join(HERE, "shaders_v1.npz"), **out)
    print("wrote", len(out), "arrays;", "wgsl steps", int(out["wgsl_steps"].sum()))


if __name__ == "__main__":
    main()
