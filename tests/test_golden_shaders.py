"""tests/golden/shaders_v1.npz: the f32 shader / post / read-out restatements are pinned
against the committed vectors on CPU; on the GPU box the HIP kernels are compared with the
same vectors (no oracle call needed)."""
import os
import sys

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
import make_golden_shaders as G  # noqa: E402  (case definitions only)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "shaders_v1.npz"))


def _same_frame(rgba, steps, g_rgba, g_steps, exact):
    if exact:   # specified f32 functions: the restatement, and the shader-order kernels, reproduce the bits
        assert np.array_equal(steps, g_steps)
        assert np.array_equal(rgba, g_rgba, equal_nan=True)
    else:       # FAST kernels: rounding differences over hundreds of f32 steps, the statistical tolerance
        ds = np.abs(steps.astype(np.int64) - g_steps.astype(np.int64))
        peak = max(float(np.abs(g_rgba[..., :3]).max()), 1e-12)
        dc = np.abs(rgba - g_rgba)[..., :3].max(-1) / peak
        assert (ds == 0).mean() >= 0.99 and (ds <= 2).mean() >= 0.995
        assert (dc <= 2e-3).mean() >= 0.99 and (dc <= 5e-2).mean() >= 0.995


def test_oracle_reproduces_shader_golden(oracle, engine_mod, gold):
    cam = engine_mod.camera_look_at(G.EYE, aspect=G.W / G.H)
    gp = engine_mod.wgsl_params(G.W, G.H, cam, 1.0, 0.999, max_steps=300, stars=0)
    gp.jitter[0], gp.jitter[1] = 0.0, -1.0 / 6.0
    rgba, steps = oracle.wgsl_frame(oracle.wgsl_params_from(gp), nthreads=4)
    _same_frame(rgba, steps, gold["wgsl_rgba"], gold["wgsl_steps"], True)
    for name in G.GLSL_CASES:
        rgba, steps = oracle.glsl_frame(oracle.glsl_params_from(G.glsl_case(name)), nthreads=4)
        _same_frame(rgba, steps, gold["glsl_%s_rgba" % name], gold["glsl_%s_steps" % name], True)
    assert np.array_equal(oracle.seeded_noise_rgba8(1)[:64], gold["noise_probe"])


def test_oracle_reproduces_post_and_readout_golden(oracle, gold):
    cur, hist = G.post_image(7), G.post_image(8)
    # every function on these paths is IEEE arithmetic or a specified routine of ref_libm.c: exact
    assert np.array_equal(oracle.taa_resolve(cur, hist, 0.75, False, True), gold["post_taa"])
    assert np.array_equal(oracle.taa_resolve(cur, hist, 0.75, True, True), gold["post_taa_moving"])
    assert np.array_equal(oracle.bloom(G.post_image(9, hdr=6.0), 0.8, 0.5, 2, True), gold["post_bloom"])
    for kind, key in ((0, "viz_curvature"), (1, "viz_tilt"), (2, "viz_frame_drag")):
        assert np.array_equal(oracle.scalar_field(kind, 1.0, 0.9, 2.2, 40.0, 9, 7), gold[key])
    assert np.array_equal(oracle.embedding_mesh(1.0, 0.9, 2.5, 30.0, 8, 6), gold["viz_embedding"])
    assert np.array_equal(oracle.ergosphere_mesh(1.0, 0.9, 9, 6), gold["viz_ergosphere"])


@pytest.mark.gpu
@pytest.mark.parametrize("arith", [0, 1])
def test_shader_kernels_against_golden(engine_mod, gold, arith):
    import torch
    n = G.W * G.H
    rgba = torch.zeros(n, 4, dtype=torch.float32, device="cuda:0")
    steps = torch.zeros(n, dtype=torch.int32, device="cuda:0")

    def grab():
        torch.cuda.synchronize()
        return rgba.cpu().numpy().reshape(G.H, G.W, 4), steps.cpu().numpy().reshape(G.H, G.W).astype(np.uint32)
    with engine_mod.PhysicsEngine(1.0, 0.9) as e:
        cam = engine_mod.camera_look_at(G.EYE, aspect=G.W / G.H)
        for wa in ((arith,) if arith == 0 else (1, 2)):     # 2: FAST with two rays per lane
            gp = engine_mod.wgsl_params(G.W, G.H, cam, 1.0, 0.999, max_steps=300, arith=wa, stars=0)
            gp.jitter[0], gp.jitter[1] = 0.0, -1.0 / 6.0
            e.render_frame_wgsl(gp, rgba, steps)
            _same_frame(*grab(), gold["wgsl_rgba"], gold["wgsl_steps"], wa == 0)
        for name in G.GLSL_CASES:
            p = G.glsl_case(name)
            p.arith = arith
            e.render_frame_glsl(p, rgba, steps)
            _same_frame(*grab(), gold["glsl_%s_rgba" % name], gold["glsl_%s_steps" % name], arith == 0)


@pytest.mark.gpu
def test_post_and_readouts_against_golden(engine_mod, gold):
    import torch
    cur, hist = torch.from_numpy(G.post_image(7)).cuda(), torch.from_numpy(G.post_image(8)).cuda()
    out = torch.zeros_like(cur)
    h, w = cur.shape[:2]

    def close(got, ref):   # shader-order kernels against the committed vectors: the same bits
        assert np.array_equal(got, ref, equal_nan=True)
    with engine_mod.PhysicsEngine(1.0, 0.9) as e:
        e.post_taa_resolve(w, h, cur, hist, out)
        torch.cuda.synchronize()
        close(out.cpu().numpy(), gold["post_taa"])
        e.post_taa_resolve(w, h, cur, hist, out, camera_moving=True)
        torch.cuda.synchronize()
        close(out.cpu().numpy(), gold["post_taa_moving"])
        scene = torch.from_numpy(G.post_image(9, hdr=6.0)).cuda()
        e.post_bloom(w, h, scene, out)
        torch.cuda.synchronize()
        close(out.cpu().numpy(), gold["post_bloom"])
        for fn, key in ((e.generate_curvature_field, "viz_curvature"), (e.generate_tilt_field, "viz_tilt"),
                        (e.generate_frame_drag_field, "viz_frame_drag")):
            assert np.array_equal(fn(2.2, 40.0, 9, 7), gold[key])
        assert np.array_equal(e.generate_embedding_mesh(2.5, 30.0, 8, 6), gold["viz_embedding"])
        assert np.array_equal(e.generate_ergosphere_mesh(9, 6), gold["viz_ergosphere"])
