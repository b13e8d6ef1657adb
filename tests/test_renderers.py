"""The reference's two renderers -- the callers of the path -- as engine calls:
grv_webgpu_render = WebGPURenderer.render (src/rendering/webgpu/renderer.ts:280-411),
grv_webgl_render  = WebGLRenderer.render's scene + post chain (src/rendering/webgl/renderer.ts:173-422).
The expected frames are composed here from the oracle's pieces in the order the renderers issue
their passes; the engine owns the history ping-pong and the frame counter."""
import numpy as np
import pytest

W, H = 96, 54


def half(x):
    return x.astype(np.float16).astype(np.float32)   # RNE, as test_post_chain pins


def halton(index, base):
    result, f, i = np.float32(0.0), np.float32(1.0) / np.float32(base), index
    while i > 0:
        result = np.float32(result + f * np.float32(i % base))
        i //= base
        f = np.float32(f / np.float32(base))
    return result


def camera_block(engine_mod, eye, prev_eye):
    """The 352-byte CameraUniforms block writeCameraUniforms would fill (types/webgpu.ts:95-116)."""
    def mats(e):
        c = engine_mod.camera_look_at(e, aspect=W / H)
        iv = np.array(c.inv_view, np.float64).reshape(4, 4).T
        ip = np.array(c.inv_proj, np.float64).reshape(4, 4).T
        return iv, ip
    iv, ip = mats(eye)
    piv, pip_ = mats(prev_eye)
    view, proj = np.linalg.inv(iv), np.linalg.inv(ip)
    prev_vp = np.linalg.inv(pip_) @ np.linalg.inv(piv)
    cu = np.zeros(88, np.float32)
    for k, m in enumerate((view, proj, iv, ip, prev_vp)):
        cu[16 * k:16 * k + 16] = m.T.reshape(-1)   # column-major
    cu[80:83] = eye
    d = -np.asarray(eye) / np.linalg.norm(eye)
    cu[84:87] = d
    return cu


def physics_block(mass, spin, frame_index=12345):
    pp = np.zeros(8, np.float32)
    pp[0], pp[1], pp[2], pp[3], pp[4], pp[5] = mass, spin, W, H, 0.0, 0.016
    pp.view(np.uint32)[6] = frame_index    # overridden by the renderer's own counter
    return pp


def test_halton_sequence():  # compute.wgsl.ts:134-145
    assert halton(1, 2) == 0.5 and abs(halton(1, 3) - 1 / 3) < 1e-7
    assert halton(2, 2) == 0.25 and halton(3, 2) == 0.75 and abs(halton(8, 3) - (2 / 3 + 2 / 9)) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("arith", [0, 1, 2])
def test_webgpu_renderer_sequence(engine_mod, oracle, arith):
    import torch
    eyes = [(59.55, -7.31, 0.0), (59.4, -7.31, 3.0), (59.0, -7.0, 6.0)]
    screen = torch.zeros(H, W, 4, dtype=torch.float32, device="cuda:0")
    hist = [np.zeros((H, W, 4), np.float32), np.zeros((H, W, 4), np.float32)]
    hi = 0
    with engine_mod.PhysicsEngine(1.0, 0.9) as e:
        for f, eye in enumerate(eyes):
            prev = eyes[f - 1] if f else eye
            cu, pp = camera_block(engine_mod, eye, prev), physics_block(1.0, 0.9)
            e.webgpu_render(cu, pp, screen, max_steps=200, arith=arith)
            torch.cuda.synchronize()
            assert e.renderer_frame_count() == f + 1
            # expected: the renderer's passes, composed from the oracle's pieces
            gp = engine_mod.WgslParams()
            gp.width, gp.height, gp.mass, gp.spin, gp.max_steps = W, H, 1.0, 0.9, 200
            gp.stars = 1                               # the renderer's compute pass always has them
            for k in range(16):
                gp.inv_view[k], gp.inv_proj[k] = cu[32 + k], cu[48 + k]
            for k in range(3):
                gp.position[k] = cu[80 + k]
            gp.jitter[0] = halton((f % 8) + 1, 2) - np.float32(0.5)
            gp.jitter[1] = halton((f % 8) + 1, 3) - np.float32(0.5)
            rgba, _ = oracle.wgsl_frame(oracle.wgsl_params_from(gp), nthreads=4)
            cam = oracle.AtaaCamera()
            for k in range(16):
                cam.inv_view[k], cam.inv_proj[k], cam.prev_view_proj[k] = cu[32 + k], cu[48 + k], cu[64 + k]
            for k in range(3):
                cam.position[k] = cu[80 + k]
            resolved = oracle.ataa_resolve(cam, half(rgba), hist[hi], True)
            hist[1 - hi] = resolved
            hi = 1 - hi
            want = resolved.copy()
            want[..., :3] = resolved[..., :3] / (resolved[..., :3] + 1.0)
            got = screen.cpu().numpy()
            d = np.abs(got - want)[..., :3].max(-1)
            # FAST march parity is statistical (tests/test_shader_kernels.py); Reinhard maps into [0, 1)
            assert (d <= 2e-3).mean() >= 0.99 and (d <= 5e-2).mean() >= 0.995, (f, d.max())
            if arith == 0:   # the whole pass sequence in shader order: frame after frame the checker's bits
                assert np.array_equal(got, want), (f, d.max())
            assert np.all(got[..., 3] == 1.0)
        e.renderer_reset()
        assert e.renderer_frame_count() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("bloom", [True, False])
def test_webgl_renderer_sequence(engine_mod, oracle, bloom):
    import torch
    screen = torch.zeros(H, W, 4, dtype=torch.float32, device="cuda:0")
    hist = [np.zeros((H, W, 4), np.float32), np.zeros((H, W, 4), np.float32)]  # ping, pong
    write = 0
    with engine_mod.PhysicsEngine(1.0, 0.9) as e:
        for f, (t, moving) in enumerate([(0.0, False), (0.5, False), (1.0, True), (1.5, False)]):
            gp = engine_mod.glsl_params(W, H, 1.0, 0.9, max_ray_steps=200, time=t, tone_map=1)
            e.webgl_render(gp, screen, bloom=bloom, camera_moving=moving)
            torch.cuda.synchronize()
            gp.tone_map = 0                       # hasPost -> ENABLE_LINEAR_OUTPUT
            scene, _ = oracle.glsl_frame(oracle.glsl_params_from(gp), nthreads=4)
            read = hist[0] if write == 0 else hist[1]          # write index 0: write pong, read ping
            resolved = oracle.taa_resolve(half(scene), read, 0.75, moving, True)
            hist[1 if write == 0 else 0] = resolved
            write = 1 - write
            want = oracle.bloom(resolved, 0.8, 0.5, 2, True) if bloom else oracle.bloom(resolved, 3e38, 0.0, 0, True)
            got = screen.cpu().numpy()
            d = np.abs(got - want)[..., :3].max(-1)
            assert np.array_equal(got, want), (f, d.max())   # shader order end to end: identical frames
            assert got[..., :3].min() >= 0.0 and got[..., :3].max() <= 1.0   # ACES + gamma
        assert e.renderer_frame_count() == 4


def _camera_block_sized(engine_mod, eye, prev_eye, w, h):
    def mats(e):
        c = engine_mod.camera_look_at(e, aspect=w / h)
        return (np.array(c.inv_view, np.float64).reshape(4, 4).T, np.array(c.inv_proj, np.float64).reshape(4, 4).T)
    iv, ip = mats(eye)
    piv, pip_ = mats(prev_eye)
    cu = np.zeros(88, np.float32)
    for k, m in enumerate((np.linalg.inv(iv), np.linalg.inv(ip), iv, ip, np.linalg.inv(pip_) @ np.linalg.inv(piv))):
        cu[16 * k:16 * k + 16] = m.T.reshape(-1)
    cu[80:83] = eye
    return cu


@pytest.mark.gpu
def test_renderers_across_a_resize(engine_mod, oracle):
    """A resolution change recreates the (zeroed) history textures but keeps the renderer's frame
    counter and history index (webgpu/renderer.ts:269-278, reprojection.ts:102-117): the frame after
    the resize blends against black and carries on the Halton jitter sequence.  Shader order: the
    composed oracle frames bit for bit."""
    import torch
    sizes = [(96, 54), (96, 54), (70, 40), (70, 40), (128, 30)]
    eyes = [(59.55, -7.31, 0.0), (59.4, -7.31, 3.0), (59.0, -7.0, 6.0), (58.5, -6.5, 9.0), (58.0, -6.0, 12.0)]
    with engine_mod.PhysicsEngine(1.0, 0.9) as e:
        hist, hi, prev_size = None, 0, None
        for f, ((w, h), eye) in enumerate(zip(sizes, eyes)):
            if (w, h) != prev_size:
                hist = [np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)]   # recreated textures
                prev_size = (w, h)
            cu = _camera_block_sized(engine_mod, eye, eyes[f - 1] if f else eye, w, h)
            pp = physics_block(1.0, 0.9)
            pp[2], pp[3] = w, h
            screen = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0")
            e.webgpu_render(cu, pp, screen, max_steps=120, arith=0)
            torch.cuda.synchronize()
            assert e.renderer_frame_count() == f + 1                   # never reset by a resize
            gp = engine_mod.WgslParams()
            gp.width, gp.height, gp.mass, gp.spin, gp.max_steps, gp.stars = w, h, 1.0, 0.9, 120, 1
            for k in range(16):
                gp.inv_view[k], gp.inv_proj[k] = cu[32 + k], cu[48 + k]
            for k in range(3):
                gp.position[k] = cu[80 + k]
            gp.jitter[0] = halton((f % 8) + 1, 2) - np.float32(0.5)
            gp.jitter[1] = halton((f % 8) + 1, 3) - np.float32(0.5)
            rgba, _ = oracle.wgsl_frame(oracle.wgsl_params_from(gp), nthreads=4)
            cam = oracle.AtaaCamera()
            for k in range(16):
                cam.inv_view[k], cam.inv_proj[k], cam.prev_view_proj[k] = cu[32 + k], cu[48 + k], cu[64 + k]
            for k in range(3):
                cam.position[k] = cu[80 + k]
            resolved = oracle.ataa_resolve(cam, half(rgba), hist[hi], True)
            hist[1 - hi] = resolved
            hi = 1 - hi
            want = resolved.copy()
            want[..., :3] = resolved[..., :3] / (resolved[..., :3] + 1.0)
            assert np.array_equal(screen.cpu().numpy(), want), (f, w, h)
    # WebGL chain: same rule for the reprojection ping-pong
    with engine_mod.PhysicsEngine(1.0, 0.9) as e:
        hist, write, prev_size = None, 0, None
        for f, (w, h) in enumerate(sizes):
            if (w, h) != prev_size:
                hist = [np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)]
                prev_size = (w, h)
            gp = engine_mod.glsl_params(w, h, 1.0, 0.9, max_ray_steps=120, time=0.5 * f, tone_map=1)
            screen = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0")
            e.webgl_render(gp, screen, bloom=True, camera_moving=False)
            torch.cuda.synchronize()
            gp.tone_map = 0
            scene, _ = oracle.glsl_frame(oracle.glsl_params_from(gp), nthreads=4)
            read = hist[0] if write == 0 else hist[1]
            resolved = oracle.taa_resolve(half(scene), read, 0.75, False, True)
            hist[1 if write == 0 else 0] = resolved
            write = 1 - write
            want = oracle.bloom(resolved, 0.8, 0.5, 2, True)
            assert np.array_equal(screen.cpu().numpy(), want), (f, w, h)
        assert e.renderer_frame_count() == len(sizes)
