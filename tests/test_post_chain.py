"""Post chain: TAA resolve (reprojection.glsl.ts), ATAA resolve (ataa.wgsl.ts),
bloom (bloom.glsl.ts + bloom.ts pass sequence).  The reference never executes these shaders in
a test (mock WebGL), so the oracle is pinned by closed-form properties on
CPU; the GPU-marked tests compare the HIP kernels with the oracle.

Stated tolerance: the kernels evaluate the same f32 expressions in the same order
(-ffp-contract=off); sqrtf/powf differ from glibc by <= 1 ulp, and a 1-ulp difference that
straddles a binary16 rounding boundary moves a stored channel by one half-ulp (2^-11
relative).  So: |d| <= 1e-3 * max(1, |ref|) everywhere, and <= 1e-6 relative on >= 99.9 %."""
import numpy as np
import pytest

RNG = np.random.default_rng(20240611)


def _image(h, w, hdr=4.0):
    img = RNG.random((h, w, 4), dtype=np.float32) ** 3 * hdr
    img[..., 3] = 1.0
    return img


def _smooth(h, w):
    """A gentle gradient: every pixel sits inside its own 3x3 variance box."""
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([0.2 + 0.01 * x, 0.5 + 0.005 * y, 0.3 + 0.002 * (x + y), np.ones_like(x)], -1)
    return np.ascontiguousarray(img, np.float32)


def _cam(oracle, engine_mod, eye, prev_eye, aspect):
    """CameraUniforms the WebGPU canvas would upload for two consecutive frames."""
    def mats(e):
        c = engine_mod.camera_look_at(e, aspect=aspect)
        inv_view = np.array(c.inv_view, np.float64).reshape(4, 4).T
        inv_proj = np.array(c.inv_proj, np.float64).reshape(4, 4).T
        return inv_view, inv_proj
    iv, ip = mats(eye)
    pv, pp = mats(prev_eye)
    prev_view_proj = np.linalg.inv(pp) @ np.linalg.inv(pv)
    cam = oracle.AtaaCamera()
    for name, m in (("inv_view", iv), ("inv_proj", ip), ("prev_view_proj", prev_view_proj)):
        flat = m.T.reshape(-1)  # column-major
        for k in range(16):
            getattr(cam, name)[k] = float(flat[k])
    for k in range(3):
        cam.position[k] = float(eye[k])
    return cam


# ---- oracle pins (CPU) --------------------------------------------------------------------
def test_round_to_half_is_binary16_rne(oracle):
    L = oracle.lib()
    xs = np.concatenate([RNG.standard_normal(4000).astype(np.float32) * 10.0 ** RNG.integers(-9, 5, 4000),
                         np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, 6.1e-5, 5.96e-8, 2.9e-8, 3.0e-8,
                                   1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11], np.float32)])
    want = xs.astype(np.float16).astype(np.float32)  # numpy converts with RNE
    got = np.array([L.orc_round_to_half(float(x)) for x in xs], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_taa_properties(oracle):
    cur = _image(24, 40)
    flat = np.full((24, 40, 4), 0.37, np.float32)
    # a flat frame has zero variance: the box collapses onto the frame, history is clipped away
    out = oracle.taa_resolve(flat, _image(24, 40), half_storage=False)
    assert np.allclose(out[..., :3], 0.37, atol=1e-6) and np.all(out[..., 3] == 1.0)
    # camera moving -> alpha = 0 -> the current frame passes through (YCoCg round trip)
    out = oracle.taa_resolve(cur, _image(24, 40), camera_moving=True, half_storage=False)
    assert np.allclose(out[..., :3], cur[..., :3], rtol=1e-5, atol=1e-6)
    # history == current on a smooth frame (pixel inside its own variance box) -> fixed point
    sm = _smooth(24, 40)
    out = oracle.taa_resolve(sm, sm, half_storage=False)
    assert np.allclose(out[1:-1, 1:-1, :3], sm[1:-1, 1:-1, :3], rtol=1e-5, atol=1e-6)
    # half storage: every stored channel is a binary16 value
    out = oracle.taa_resolve(cur, _image(24, 40))
    assert np.array_equal(out[..., :3], out[..., :3].astype(np.float16).astype(np.float32))
    # a darker history is clipped to the lower box face: the blend never brightens, and never
    # leaves the 1.5 sigma neighbourhood (anti-ghosting clamp, reprojection.glsl.ts:93-99)
    out = oracle.taa_resolve(sm, sm * 0.5, blend_factor=0.9, half_storage=False)
    y = lambda im: im[..., 0] * 0.25 + im[..., 1] * 0.5 + im[..., 2] * 0.25  # noqa: E731
    inner = (slice(1, -1), slice(1, -1))
    assert np.all(y(out)[inner] <= y(sm)[inner] + 1e-6)
    assert np.all(y(out)[inner] >= y(sm)[inner] - 0.02)


def test_ataa_static_camera_is_a_fixed_point(oracle, engine_mod):
    eye = (59.55, -7.31, 0.0)
    cam = _cam(oracle, engine_mod, eye, eye, 40 / 24)
    cur = _smooth(24, 40)
    out = oracle.ataa_resolve(cam, cur, cur, half_storage=False)
    # prev_view_proj == view_proj: prevUV == uv, so history is sampled at texel centres
    assert np.allclose(out[1:-1, 1:-1, :3], cur[1:-1, 1:-1, :3], rtol=2e-4, atol=2e-5)
    moved = _cam(oracle, engine_mod, eye, (59.0, -7.31, 4.0), 40 / 24)
    out2 = oracle.ataa_resolve(moved, cur, _image(24, 40), half_storage=False)
    assert np.all(np.isfinite(out2)) and not np.allclose(out2, out)


def test_bloom_properties(oracle):
    dark = _image(32, 48, hdr=0.5)
    dark[..., :3] = np.minimum(dark[..., :3], 0.7)   # luminance below the 0.8 threshold
    aces = lambda c: np.clip((c * (2.51 * c + 0.03)) / (c * (2.43 * c + 0.59) + 0.14), 0, 1)  # noqa: E731
    out = oracle.bloom(dark)
    # nothing passes the bright pass: combine reduces to ACES + gamma of the scene
    assert np.allclose(out[..., :3], aces(dark[..., :3].astype(np.float64)) ** 0.4545, atol=2e-6)
    # a uniformly bright scene: the Gaussian weights sum to 1.0000004 per pass (bloom.glsl.ts:74)
    bright = np.full((32, 48, 4), 2.0, np.float32)
    out = oracle.bloom(bright, intensity=0.5, blur_passes=2, half_storage=False)
    wsum = 0.227027 + 2 * (0.1945946 + 0.1216216 + 0.054054 + 0.016216)
    want = aces(2.0 + 2.0 * wsum ** 4 * 0.5) ** 0.4545
    assert np.allclose(out[..., :3], want, atol=1e-5)
    # zero passes: the combine samples the half-resolution bright texture directly (bloom.ts:512-540)
    out0 = oracle.bloom(bright, blur_passes=0, half_storage=False)
    assert np.allclose(out0[..., :3], aces(2.0 + 2.0 * 0.5) ** 0.4545, atol=1e-5)
    # bloom only adds light
    sc = _image(32, 48, hdr=6.0)
    assert np.all(oracle.bloom(sc)[..., :3] >= oracle.bloom(sc, intensity=0.0)[..., :3] - 1e-6)
    # odd sizes: floor(w/2), floor(w/4), at least 1 (bloom.ts:473-474, 509-510)
    tiny = oracle.bloom(_image(3, 5, hdr=6.0))
    assert tiny.shape == (3, 5, 4) and np.all(np.isfinite(tiny))


def test_ataa_reprojection_fold_is_the_shader_chain(engine_mod, oracle):
    """CPU: the 24 coefficients the FAST ATAA resolve evaluates its history tap from
    (grv_ataa_reproj_fold: the launcher's f64 fold of ataa.wgsl.ts:60-75) against the chain itself
    evaluated in f64 -- they must describe the same tap to f32 rounding of the coefficients, and the fold
    must sit at least as close to that position as the shader-order f32 chain does."""
    import ctypes as C
    bh = engine_mod
    L = bh.load_library()
    f = np.float32
    for (h, w), eye, prev in (((540, 960), (59.55, -7.31, 0.0), (59.0, -7.31, 4.0)),
                              ((2160, 3840), (20.0, 3.0, -35.0), (20.5, 3.2, -34.0)),
                              ((37, 53), (0.0, 1.0, 60.0), (0.0, 1.0, 60.0))):  # last: static camera
        cam = _cam(oracle, bh, eye, prev, w / h)
        ap = bh.AtaaParams()
        ap.width, ap.height = w, h
        for name in ("inv_view", "inv_proj", "prev_view_proj"):
            for k in range(16):
                getattr(ap, name)[k] = getattr(cam, name)[k]
        for k in range(3):
            ap.position[k] = cam.position[k]
        out = (C.c_float * 24)()
        assert L.grv_ataa_reproj_fold(C.byref(ap), out) == 0
        co = np.array(out[:21], np.float64).reshape(7, 3)
        kk = np.array(out[21:], np.float64)
        IV = np.array(ap.inv_view[:], np.float64).reshape(4, 4).T
        IP = np.array(ap.inv_proj[:], np.float64).reshape(4, 4).T
        PVP = np.array(ap.prev_view_proj[:], np.float64).reshape(4, 4).T
        pos = np.array(ap.position[:], np.float64)
        step = max(1, w // 480)  # every pixel of the small frames, a 1/64 lattice of the 4K one
        px, py = np.meshgrid(np.arange(0, w, step, dtype=np.float64), np.arange(0, h, step, dtype=np.float64))

        def chain(dt):
            n = np.stack([((px + 0.5) / w * 2 - 1).astype(dt), (-((py + 0.5) / h * 2 - 1)).astype(dt),
                          np.ones_like(px, dt), np.ones_like(px, dt)], -1)
            vt = n @ IP.astype(dt).T
            v = vt[..., :3] / vt[..., 3:4]
            v = v / np.sqrt((v * v).sum(-1, keepdims=True)).astype(dt)
            world = pos.astype(dt) + (v @ IV.astype(dt)[:3, :3].T) * dt(12)
            pc = np.concatenate([world, np.ones_like(world[..., :1])], -1) @ PVP.astype(dt).T
            pu = (pc[..., 0] / pc[..., 3]) * dt(.5) + dt(.5)
            pv = (pc[..., 1] / pc[..., 3]) * dt(-.5) + dt(.5)
            return (pu * dt(w) - dt(.5)).astype(np.float64), (pv * dt(h) - dt(.5)).astype(np.float64)
        xe, ye = chain(np.float64)
        x32, y32 = chain(f)
        q = [co[i, 0] * px + co[i, 1] * py + co[i, 2] for i in range(7)]
        s = 12.0 / np.sqrt(q[0] ** 2 + q[1] ** 2 + q[2] ** 2) * np.sign(q[3])
        ipw = 1.0 / (kk[2] + s * q[6])
        xf, yf = (kk[0] + s * q[4]) * ipw, (kk[1] + s * q[5]) * ipw
        d_fold = np.hypot(xf - xe, yf - ye)
        d_f32 = np.hypot(x32 - xe, y32 - ye)
        # the fold's only error is the f32 rounding of its 24 coefficients: a few ulps of the texel
        # coordinate, and no worse than the shader-order chain's own rounding
        assert d_fold.max() <= 8 * 2.0 ** -23 * max(w, h), (d_fold.max(), w, h)
        assert np.median(d_fold) <= 1.5 * np.median(d_f32) + 1e-7, (np.median(d_fold), np.median(d_f32))
    assert L.grv_ataa_reproj_fold(None, out) != 0


# ---- HIP kernels vs oracle (GPU box) ----------------------------------------------------------
def _close(got, ref, fast=False, tap_noise=0.0):
    d = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    assert d.max() <= 1e-3 + tap_noise, d.max()
    rel = np.abs(got - ref) / np.maximum(1e-6, np.abs(ref))
    if fast:
        # FMA, reciprocal-based divide / sqrt, exp2-log2 gamma.  The approximate divide also moves a
        # texture coordinate by an ulp, i.e. the tap by ~1e-5 of a texel, which (with f32 bilinear
        # weights) leaks that fraction of the neighbouring texel into the tap: O(1e-5) absolute on
        # O(1) HDR values; the binary16 store then adds at most one half-ulp (2^-11 relative).
        ad = np.abs(got - ref)
        ok = ad <= 1e-4 + tap_noise + 1e-3 * np.abs(ref)
        # (a reprojected tap -- tap_noise > 0 -- also moves the median: the checker's own f32 chain sits a
        # median 6e-5 texel from the exactly evaluated position on the 960-wide case, the folded FAST form
        # 4e-5; 1/32 of the worst-case bound covers the two)
        assert ok.mean() >= 0.999 and np.median(ad) <= 1e-5 + tap_noise / 32.0, (ok.mean(), np.median(ad))
    else:   # shader order, IEEE divide / sqrt, specified powf: the checker's bits
        assert np.array_equal(got, ref, equal_nan=True), (rel <= 1e-6).mean()


@pytest.mark.gpu
@pytest.mark.parametrize("half", [True, False])
@pytest.mark.parametrize("size", [(135, 240), (37, 53), (136, 240), (16, 64), (17, 65), (5, 3), (540, 960)])
def test_fast_post_chain_matches_oracle(engine_mod, oracle, half, size):
    """The FAST contract of the post kernels (kernels_fast.hip) against the same oracle."""
    import torch
    h, w = size
    cur, hist = _image(h, w), _image(h, w)
    dc, dh = torch.from_numpy(cur).cuda(), torch.from_numpy(hist).cuda()
    out = torch.zeros_like(dc)
    with engine_mod.PhysicsEngine(1.0, 0.9) as e:
        e.post_taa_resolve(w, h, dc, dh, out, half_storage=half, arith=engine_mod.ARITH_FAST)
        torch.cuda.synchronize()
        _close(out.cpu().numpy(), oracle.taa_resolve(cur, hist, 0.75, False, half), fast=True)
        cam = _cam(oracle, engine_mod, (59.55, -7.31, 0.0), (59.0, -7.31, 4.0), w / h)
        ap = engine_mod.AtaaParams()
        ap.width, ap.height, ap.half_storage, ap.arith = w, h, 1 if half else 0, engine_mod.ARITH_FAST
        for name in ("inv_view", "inv_proj", "prev_view_proj"):
            for k in range(16):
                getattr(ap, name)[k] = getattr(cam, name)[k]
        for k in range(3):
            ap.position[k] = cam.position[k]
        e.post_ataa_resolve(ap, dc, dh, out)
        torch.cuda.synchronize()
        # the reprojected history tap: the FAST kernel evaluates the chain with its matrices folded (in
        # f64, by the launcher) and one rsq + one rcp, the checker in the shader's f32 order with three
        # 4x4 products, two divides and a square root -- the two coordinates differ by the CHECKER's own
        # rounding, a few f32 ulps of uv, i.e. ~4 * 2^-23 * max(w, h) texels (the folded form is the one
        # closer to the exactly evaluated position), times the texel-to-texel step of this noise image
        # (up to its full range, 4) -- a bound that grows with the image, unlike the others
        _close(out.cpu().numpy(), oracle.ataa_resolve(cam, cur, hist, half), fast=True,
               tap_noise=4.0 * 2.0 ** -23 * max(w, h) * 4.0)
        scene = _image(h, w, hdr=6.0)
        ds = torch.from_numpy(scene).cuda()
        e.post_bloom(w, h, ds, out, half_storage=1 if half else 0, arith=engine_mod.ARITH_FAST)
        torch.cuda.synchronize()
        _close(out.cpu().numpy(), oracle.bloom(scene, 0.8, 0.5, 2, half), fast=True)


@pytest.mark.gpu
@pytest.mark.parametrize("half", [True, False])
@pytest.mark.parametrize("size,passes", [((128, 256), 2), ((64, 64), 1), ((36, 52), 3), ((4, 8), 1), ((8, 4), 2),
                                         ((136, 200), 4), ((540, 960), 2), ((60, 224), 2), ((135, 240), 2), ((3, 5), 1)])
def test_fast_bloom_forms_match_oracle(engine_mod, oracle, half, size, passes):
    """The FAST bloom launches (post_fast_kernels.hpp: 56-wide stages, texel reads) on nested sizes of
    every tile remainder, and the unfused FAST fallback on sizes that do not nest."""
    import torch
    h, w = size
    scene = _image(h, w, hdr=6.0)
    ds = torch.from_numpy(scene).cuda()
    out = torch.zeros_like(ds)
    with engine_mod.PhysicsEngine(1.0, 0.9) as e:
        e.post_bloom(w, h, ds, out, blur_passes=passes, half_storage=1 if half else 0, arith=engine_mod.ARITH_FAST)
        torch.cuda.synchronize()
    _close(out.cpu().numpy(), oracle.bloom(scene, 0.8, 0.5, passes, half), fast=True)


@pytest.mark.gpu
@pytest.mark.parametrize("half", [True, False])
@pytest.mark.parametrize("size", [(135, 240), (37, 53)])
def test_taa_and_ataa_kernels_match_oracle(engine_mod, oracle, half, size):
    import torch
    h, w = size
    cur, hist = _image(h, w), _image(h, w)
    dc, dh = torch.from_numpy(cur).cuda(), torch.from_numpy(hist).cuda()
    out = torch.zeros_like(dc)
    with engine_mod.PhysicsEngine(1.0, 0.9) as e:
        for moving in (False, True):
            e.post_taa_resolve(w, h, dc, dh, out, blend_factor=0.75, camera_moving=moving, half_storage=half)
            torch.cuda.synchronize()
            _close(out.cpu().numpy(), oracle.taa_resolve(cur, hist, 0.75, moving, half))
        cam = _cam(oracle, engine_mod, (59.55, -7.31, 0.0), (59.0, -7.31, 4.0), w / h)
        ap = engine_mod.AtaaParams()
        ap.width, ap.height, ap.half_storage = w, h, 1 if half else 0
        for name in ("inv_view", "inv_proj", "prev_view_proj"):
            for k in range(16):
                getattr(ap, name)[k] = getattr(cam, name)[k]
        for k in range(3):
            ap.position[k] = cam.position[k]
        e.post_ataa_resolve(ap, dc, dh, out)
        torch.cuda.synchronize()
        _close(out.cpu().numpy(), oracle.ataa_resolve(cam, cur, hist, half))
        # aliasing an input with the output is refused (the stencil reads neighbours)
        with pytest.raises(engine_mod.GravitasError):
            e.post_taa_resolve(w, h, dc, dh, dc)


@pytest.mark.gpu
@pytest.mark.parametrize("half", [True, False])
@pytest.mark.parametrize("size,passes", [((135, 240), 2), ((64, 64), 0), ((37, 53), 3), ((3, 5), 1),
                                         # w, h multiples of 4: the fused launches (bright + H, (V, H) pairs, V + combine)
                                         ((128, 256), 2), ((64, 64), 1), ((36, 52), 3), ((4, 8), 1), ((8, 4), 2),
                                         ((136, 200), 4), ((540, 960), 2)])
def test_bloom_kernels_match_oracle(engine_mod, oracle, half, size, passes):
    import torch
    h, w = size
    scene = _image(h, w, hdr=6.0)
    ds = torch.from_numpy(scene).cuda()
    out = torch.zeros_like(ds)
    with engine_mod.PhysicsEngine(1.0, 0.9) as e:
        e.post_bloom(w, h, ds, out, blur_passes=passes, half_storage=1 if half else 0)
        torch.cuda.synchronize()
    _close(out.cpu().numpy(), oracle.bloom(scene, 0.8, 0.5, passes, half))


@pytest.mark.gpu
def test_half_rounding_on_device_is_the_oracle_rounding(engine_mod, oracle):
    """__float2half_rn / __half2float (post_store) against the oracle's software RNE, through
    the TAA kernel with camera_moving (pass-through of the current frame, then the store)."""
    import torch
    h, w = 64, 256
    cur = (RNG.standard_normal((h, w, 4)).astype(np.float32) * 10.0 ** RNG.integers(-6, 4, (h, w, 1))).astype(np.float32)
    cur[..., 1] = cur[..., 0]
    cur[..., 2] = cur[..., 0]     # grey: YCoCg round trip is exact up to the adds
    dc = torch.from_numpy(cur).cuda()
    out = torch.zeros_like(dc)
    with engine_mod.PhysicsEngine(1.0, 0.0) as e:
        e.post_taa_resolve(w, h, dc, dc, out, camera_moving=True, half_storage=True)
        torch.cuda.synchronize()
    got = out.cpu().numpy()
    ref = oracle.taa_resolve(cur, cur, 0.75, True, True)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
