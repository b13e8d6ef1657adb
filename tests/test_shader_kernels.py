"""f32 march loops of the reference's shaders: the CPU restatement is
sanity-checked on CPU; on the GPU box the HIP kernels are compared with it.

The shader-order (STRICT) kernels evaluate the shaders' transcendental functions by the same
specified f32 forms as the restatement (csrc/strict_libm.hpp <-> oracle/ref_libm.c) and must return
its step counts and pixels bit for bit, stars included.  The FAST kernels (FMA, hardware
rcp/rsq/exp2/log2, polynomial sin/cos) amplify their rounding differences over hundreds of f32
steps, so their pixel parity is statistical: the stated tolerance is FAST_BARS below (what is measured, plus a bounded tail), and BASELINE configs[3] is checked in the very form
the bench runs it (test_config4_bench_form_against_the_oracle)."""
import ctypes as C

import numpy as np
import pytest

EYE = (60.0 * np.sin(np.deg2rad(97.0)), 60.0 * np.cos(np.deg2rad(97.0)), 0.0)


def test_wgsl_oracle_frame_is_sane(oracle, engine_mod):
    W, H = 96, 54
    cam = engine_mod.camera_look_at(EYE, aspect=W / H)
    gp = engine_mod.wgsl_params(W, H, cam, 1.0, 0.9, max_steps=512, stars=0)
    rgba, steps = oracle.wgsl_frame(oracle.wgsl_params_from(gp), nthreads=4)
    assert rgba.shape == (H, W, 4) and np.all(rgba[..., 3] == 1.0) and np.all(np.isfinite(rgba))
    assert steps.max() <= 512 and steps.min() >= 1
    lit = rgba[..., :3].sum(-1) > 0
    assert 0.02 < lit.mean() < 0.6               # the disk is visible, the sky is black
    # the image of an equatorial disk seen from 97 deg is brighter on the approaching side
    assert rgba[..., 0].max() > 0.1


def test_wgsl_star_hash_density(oracle, engine_mod):
    """compute.wgsl.ts:199-206: fract(sin(dot(v, k)) * 43758.5453) > 0.999 lights ~0.1 % of the
    escaping rays.  The hash turns the last ulp of its argument into O(1), so the star *pattern*
    cannot be compared between implementations -- its density can."""
    W, H = 256, 144
    cam = engine_mod.camera_look_at(EYE, aspect=W / H)
    on = engine_mod.wgsl_params(W, H, cam, 1.0, 0.9, max_steps=512)
    assert on.stars == 1                                        # the shader always has them
    off = engine_mod.wgsl_params(W, H, cam, 1.0, 0.9, max_steps=512, stars=0)
    a, sa = oracle.wgsl_frame(oracle.wgsl_params_from(on), nthreads=4)
    b, sb = oracle.wgsl_frame(oracle.wgsl_params_from(off), nthreads=4)
    assert np.array_equal(sa, sb)
    star = (a[..., 0] - b[..., 0]) > 0.5
    dark = b[..., :3].sum(-1) == 0
    assert not (star & ~dark).any() or (a[star & ~dark] <= b[star & ~dark] + 1.0).all()
    frac = star.sum() / max(dark.sum(), 1)
    assert 2e-4 < frac < 4e-3, frac


def test_glsl_oracle_frame_is_sane(oracle, engine_mod):
    W, H = 96, 54
    # march + disk only (no stars/nebula): the sky is black, the disk is lit
    gp = engine_mod.glsl_params(W, H, 1.0, 0.9, max_ray_steps=256, features=7, turbulence=0.75)
    rgba, steps = oracle.glsl_frame(oracle.glsl_params_from(gp), nthreads=4)
    assert np.all(np.isfinite(rgba)) and np.all(rgba[..., 3] == 1.0)
    assert steps.max() <= 256
    lit = rgba[..., :3].sum(-1) > 0
    assert 0.02 < lit.mean() < 0.9
    # reference default preset: the nebula term lights every escaping ray
    gpd = engine_mod.glsl_params(W, H, 1.0, 0.9, max_ray_steps=256)
    assert gpd.features == engine_mod.GLSL_FEATURES_DEFAULT and gpd.quality == 1
    full, _ = oracle.glsl_frame(oracle.glsl_params_from(gpd), nthreads=4)
    assert np.all(np.isfinite(full)) and (full[..., :3].sum(-1) > 0).mean() > lit.mean()
    # shader cap: u_maxRaySteps above 500 is clamped (fragment.glsl.ts:115)
    gp2 = engine_mod.glsl_params(32, 18, 1.0, 0.9, max_ray_steps=4000)
    _, st2 = oracle.glsl_frame(oracle.glsl_params_from(gp2))
    assert st2.max() <= 500
    # tone mapping maps into [0, 1]
    gp3 = engine_mod.glsl_params(32, 18, 1.0, 0.9, tone_map=1)
    r3, _ = oracle.glsl_frame(oracle.glsl_params_from(gp3))
    assert r3[..., :3].max() <= 1.0


def test_glsl_oracle_compositing_terms(oracle, engine_mod):
    """each ShaderManager #define changes exactly the term it gates."""
    bh = engine_mod
    W, H = 96, 54
    base = bh.GLSL_LENSING | bh.GLSL_DISK | bh.GLSL_DOPPLER

    def frame(**kw):
        gp = bh.glsl_params(W, H, 1.0, 0.9, max_ray_steps=256, **kw)
        return oracle.glsl_frame(oracle.glsl_params_from(gp), nthreads=4)
    plain, st_plain = frame(features=base, turbulence=0.75)
    # the seeded textures agree between the product helper and the oracle helper
    assert np.array_equal(bh.seeded_noise_rgba8(1), oracle.seeded_noise_rgba8(1))
    assert bh.seeded_noise_rgba8(2).max() <= 254            # floor(u * 255), webgl-utils.ts:264-266
    # stars: the march is unchanged; rays that saw no disk gain the sky.  (Behind the disk the
    # term is background * (1 - alpha) and the shader lets alpha pass 1: fragment.glsl.ts:276.)
    sky = plain[..., :3].sum(-1) == 0
    stars, st = frame(features=base | bh.GLSL_STARS, turbulence=0.75)
    assert np.array_equal(st, st_plain) and (stars[sky][:, :3] >= 0).all()
    assert (stars[sky][:, :3] > 0).any()
    # photon glow adds light to disk-free rays only; jets add emission above/below the hole
    glow, _ = frame(features=base | bh.GLSL_PHOTON_GLOW, turbulence=0.75)
    assert (glow[sky][:, :3] >= 0).all()
    jets, _ = frame(features=base | bh.GLSL_JETS, turbulence=0.75)
    assert jets[..., 2].sum() > plain[..., 2].sum()
    # jets without a disk are suppressed (manager.ts:72-73)
    nodisk, _ = frame(features=bh.GLSL_LENSING | bh.GLSL_JETS, turbulence=0.75)
    nodisk2, _ = frame(features=bh.GLSL_LENSING, turbulence=0.75)
    assert np.array_equal(nodisk, nodisk2)
    # texture turbulence modulates the disk, dither moves the start point by < MIN_STEP
    tex, _ = frame(features=base)
    assert not np.array_equal(tex, plain) and np.isfinite(tex).all()
    # redshift overlay replaces the image by the heat map; the disk is skipped (disk.ts:21)
    red, _ = frame(features=base | bh.GLSL_REDSHIFT, show_redshift=1.0, turbulence=0.75)
    assert set(np.unique(red[..., 3])) == {1.0} and red[..., :3].max() <= 1.0 + 1e-6
    assert (red[..., 1] <= red[..., 0] + red[..., 2] + 1e-6).all()
    # Kerr shadow guide paints green along the Bardeen curve
    curve = oracle.bardeen_shadow(1.0, 0.9, np.deg2rad(97.0), 32)
    guide, _ = frame(features=base, turbulence=0.75, show_kerr_shadow=1.0, shadow_curve=curve[:64])
    changed = np.abs(guide - plain)[..., :3].max(-1) > 0
    assert 0 < changed.mean() < 0.1
    assert (guide[changed][:, 1] >= plain[changed][:, 1]).all()
    # low-quality indicator path does not march; debug path shows uv
    low, st_low = frame(features=base | bh.GLSL_STARS, quality=0)
    assert st_low.max() == 0 and np.isfinite(low).all() and low[..., :3].max() > 0
    dbg, _ = frame(debug=1.0)
    assert abs(dbg[0, 0, 1] - (0.5 - 0.5 / H + 0.5)) < 1e-6 and np.all(dbg[..., 2] == 0)
    # SAB camera path (u_camPos != 0): looking from +z at the hole still finds the disk
    cam, _ = frame(features=base, turbulence=0.75, cam_pos=(0.0, 6.0, -60.0),
                   cam_quat=(np.sin(0.05), 0.0, 0.0, np.cos(0.05)))
    assert (cam[..., :3].sum(-1) > 0).mean() > 0.01


# FAST contract, stated tolerance (round 3: what is measured, plus a bounded tail).  Measured on MI355X
# (tests/measure_f32_fast_tail.py -> profiles/r03_f32_fast_tail.json; WGSL march, FAST and packed, 480x270 and
# every 16th pixel of the 7680x4320 / 1024-step frame of BASELINE configs[3]): step counts equal on
# 99.98 % of the pixels; |d colour| / peak: p50 = p90 = 0 (91 % of the pixels are bit-identical), p99
# 2.3e-7, p99.9 3.4e-6, p99.99 4e-5, max 3.9e-2; no pixel beyond 5e-2.  The pixels that differ at all
# in step count are the rays next to the critical curve (they orbit: up to the whole step budget of
# difference) -- they end in the hole or far away either way, so their colour barely moves.
FAST_BARS = dict(steps_equal=0.999, steps_within_2=0.9995, colour_1e4=0.999, colour_2e3=0.9995, beyond_5e2=1e-4)
# The GLSL fragment shader (FAST) measures tighter still on its nine cases at 480x270: step counts
# equal on >= 99.997 %, |d colour| / peak <= 1e-4 on >= 99.99 %, max 1.1e-2 -- it is held to the same bars.


def _compare(got_rgba, got_steps, ref_rgba, ref_steps, exact=False, bars=FAST_BARS):
    if exact:   # shader order with the specified f32 functions: the checker's bits
        assert np.array_equal(got_steps, ref_steps)
        assert np.array_equal(got_rgba, ref_rgba, equal_nan=True)
        return None
    ds = np.abs(got_steps.astype(np.int64) - ref_steps.astype(np.int64))
    peak = max(float(ref_rgba[..., :3].max()), 1e-12)
    dc = np.abs(got_rgba - ref_rgba)[..., :3].max(-1) / peak
    m = dict(steps_equal=float((ds == 0).mean()), steps_within_2=float((ds <= 2).mean()),
             colour_1e4=float((dc <= 1e-4).mean()), colour_2e3=float((dc <= 2e-3).mean()),
             beyond_5e2=float((dc > 5e-2).mean()), colour_max=float(dc.max()),
             colour_p99=float(np.percentile(dc, 99)), colour_p9999=float(np.percentile(dc, 99.99)),
             identical=float((got_rgba == ref_rgba).all(-1).mean()))
    print("FAST vs oracle:", m)
    for k in ("steps_equal", "steps_within_2", "colour_1e4", "colour_2e3"):
        assert m[k] >= bars[k], (k, m[k], bars[k])
    assert m["beyond_5e2"] <= bars["beyond_5e2"], m
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("arith", [0, 1, 2])
def test_config4_bench_form_against_the_oracle(engine_mod, oracle, arith):
    """BASELINE configs[3] exactly as `bench.py --config c4` runs it -- 7680x4320, fixed 1024-step
    budget, FAST contract (arith 2 = two rays per lane on the packed-f32 ops, the bench default) --
    against the shader-order oracle on every 16th pixel in x and y (129 600 rays), stars off (one ulp
    of the hash's sin lights a different star: the bench line's sky differs star by star from the
    shader-order sky, and config.workload says so).  The FAST checks otherwise stop at 480x270.
    arith 0 = the shader-order kernel: the checker's bits, stars on.  GRV_C4_STRIDE=1 compares EVERY
    pixel of the 33.2 M (the C oracle marches the 4.9 G steps of the frame in about a minute on 16
    cores); GRV_C4_JSON=<path> appends the measured figures (profiles/r03_full_frame_parity_c4.jsonl)."""
    import json
    import os
    import torch
    W, H, stride = 7680, 4320, int(os.environ.get("GRV_C4_STRIDE", "16"))
    cam = engine_mod.camera_look_at(EYE, aspect=W / H)
    gp = engine_mod.wgsl_params(W, H, cam, 1.0, 0.999, max_steps=1024, arith=arith, stars=1 if arith == 0 else 0)
    with engine_mod.PhysicsEngine(1.0, 0.999) as e:
        rgba = torch.zeros(W * H, 4, dtype=torch.float32, device="cuda:0")
        steps = torch.zeros(W * H, dtype=torch.int32, device="cuda:0")
        tot = e.render_frame_wgsl(gp, rgba, steps)
        assert tot == int(steps.sum(dtype=torch.int64).item())
        g = rgba.view(H, W, 4)[::stride, ::stride].cpu().numpy()
        s = steps.view(H, W)[::stride, ::stride].cpu().numpy()
    ref_rgba, ref_steps = oracle.wgsl_frame(oracle.wgsl_params_from(gp), stride=(stride, stride), nthreads=16)
    assert ref_steps.shape == s.shape == ((H + stride - 1) // stride, (W + stride - 1) // stride)
    m = _compare(g, s, ref_rgba, ref_steps, exact=(arith == 0))
    if arith != 0:
        # on the 129 600-pixel lattice nothing differs by more than 5e-2 of the peak; on all 33.2 M
        # pixels 348 do (1.05e-5 of them, up to 0.37 of the peak): near-critical rays whose step count
        # differs, where the one decides for the disk and the other for the hole -- bounded by
        # FAST_BARS["beyond_5e2"] = 1e-4 inside _compare
        assert stride < 16 or m["colour_max"] <= 5e-2
        # rays still marching when the budget runs out orbit next to the critical curve, where one ulp
        # decides between another turn and falling in: a handful of the 129 600 on either side, and the
        # SAME handful to within a few rays -- round 3's FAST forms kept four times as many (13 against 3
        # here, 4 159 against 1 007 on the whole frame): rays that cross the polar axis blow up in the
        # shader's spherical coordinates, and the FAST forms' NaN passed no exit test
        # (profiles/r04_c4_budget_rays.json; the exits now test !(r >= r_stop))
        scale = max(1, (16 // stride) ** 2)
        n_eng, n_ref = int((s == 1024).sum()), int((ref_steps == 1024).sum())
        assert n_eng <= 40 * scale and n_ref <= 40 * scale
        assert abs(n_eng - n_ref) <= max(4 * scale ** 0.5, 0.15 * n_ref), (n_eng, n_ref)
    path = os.environ.get("GRV_C4_JSON")
    if path:
        rec = {"frame": "7680x4320 f32 compute march, 1024-step budget, a=0.999", "arith": ["shader order", "FAST", "FAST packed"][arith],
               "stride": stride, "pixels_compared": int(s.size), "oracle_steps": int(ref_steps.sum()),
               "at_budget_engine": int((s == 1024).sum()), "at_budget_oracle": int((ref_steps == 1024).sum())}
        rec.update(m if m else {"bit_identical": True})
        with open(path, "a") as f:
            f.write(json.dumps(rec) + "\n")


@pytest.mark.gpu
@pytest.mark.parametrize("arith", [1, 2])
def test_fast_marches_leave_with_the_shader_when_a_ray_crosses_the_polar_axis(engine_mod, oracle, arith):
    """The compute march has no polar special case (compute.wgsl.ts:49 clamps sin^2 at 1e-12): a ray that
    steps across the axis blows up in every arithmetic.  Shader order (and a double evaluation of the
    same march, oracle/wgsl_f64_twin.c) lands on a huge negative r and leaves through r < 1.001 r+;
    the FAST forms land on NaN and must leave at the same point instead of marching the rest of the
    budget as NaN.  Pixels are picked by the double march: theta leaves [0, pi]."""
    import torch
    W, H, budget = 1920, 1080, 512
    cam = engine_mod.camera_look_at(EYE, aspect=W / H)
    gp = engine_mod.wgsl_params(W, H, cam, 1.0, 0.999, max_steps=budget, arith=arith, stars=0)
    op = oracle.wgsl_params_from(gp)
    ys, xs = np.mgrid[300:780, 700:1220]            # the shadow and its surroundings
    xy = np.stack([xs.ravel(), ys.ravel()], 1)
    d = oracle.wgsl_pixels_f64(op, xy, nthreads=16)
    polar = (d["axis_margin"] < 0.0) & (d["cls"] != 2)
    assert polar.sum() >= 20, int(polar.sum())
    with engine_mod.PhysicsEngine(1.0, 0.999) as e:
        rgba = torch.zeros(W * H, 4, dtype=torch.float32, device="cuda:0")
        steps = torch.zeros(W * H, dtype=torch.int32, device="cuda:0")
        e.render_frame_wgsl(gp, rgba, steps)
        got = steps.view(H, W).cpu().numpy()[xy[polar, 1], xy[polar, 0]].astype(np.int64)
    ref = np.array([oracle.lib().orc_wgsl_pixel(C.byref(op), int(x), int(y), (C.c_float * 4)()) for x, y in xy[polar]], np.int64)
    print("polar-axis rays: %d, FAST at budget %d, shader order at budget %d, max |d steps| %d"
          % (polar.sum(), (got == budget).sum(), (ref == budget).sum(), np.abs(got - ref).max()))
    assert (got == budget).sum() <= (ref == budget).sum() + 1
    assert np.percentile(np.abs(got - ref), 95) <= 3


@pytest.mark.gpu
@pytest.mark.parametrize("arith", [0, 1, 2])  # shader order / FAST / FAST with two rays per lane (packed f32)
@pytest.mark.parametrize("spin,max_steps", [(0.999, 512), (0.5, 150)])
def test_wgsl_kernel_matches_oracle(engine_mod, oracle, spin, max_steps, arith):
    import torch
    W, H = 480, 270
    cam = engine_mod.camera_look_at(EYE, aspect=W / H)
    # the star hash fract(sin(.) * 43758.5) turns one ulp of sin into a different sky: only the
    # shader-order kernel (same bits as the checker) is compared with the stars on
    gp = engine_mod.wgsl_params(W, H, cam, 1.0, spin, max_steps=max_steps, arith=arith, stars=1 if arith == 0 else 0)
    gp.jitter[0], gp.jitter[1] = 0.0, -1.0 / 6.0        # frame-0 Halton jitter
    n = W * H
    with engine_mod.PhysicsEngine(1.0, spin) as e:
        rgba = torch.zeros(n, 4, dtype=torch.float32, device="cuda:0")
        steps = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        tot = e.render_frame_wgsl(gp, rgba, steps)
    ref_rgba, ref_steps = oracle.wgsl_frame(oracle.wgsl_params_from(gp), nthreads=8)
    assert tot == int(steps.sum().item())
    _compare(rgba.cpu().numpy().reshape(H, W, 4), steps.cpu().numpy().reshape(H, W), ref_rgba, ref_steps,
             exact=(arith == 0))


@pytest.mark.gpu
@pytest.mark.parametrize("arith", [0, 1, 2])
def test_wgsl_kernel_star_density(engine_mod, arith):
    import torch
    W, H = 960, 540
    cam = engine_mod.camera_look_at(EYE, aspect=W / H)
    out = {}
    with engine_mod.PhysicsEngine(1.0, 0.9) as e:
        for stars in (1, 0):
            gp = engine_mod.wgsl_params(W, H, cam, 1.0, 0.9, max_steps=512, arith=arith, stars=stars)
            rgba = torch.zeros(W * H, 4, dtype=torch.float32, device="cuda:0")
            e.render_frame_wgsl(gp, rgba)
            torch.cuda.synchronize()
            out[stars] = rgba.cpu().numpy()
    star = (out[1][:, 0] - out[0][:, 0]) > 0.5
    dark = out[0][:, :3].sum(-1) == 0
    frac = star.sum() / dark.sum()
    assert 1e-4 < frac < 4e-3, frac        # sparse: of the order of 0.1 % of the dark sky
    assert np.array_equal(out[1][~star], out[0][~star])


@pytest.mark.gpu
@pytest.mark.parametrize("spin,tone,kw", [
    (0.999, 0, dict(features=7, turbulence=0.75)),            # a16/a17 only: march + disk Doppler
    (0.9, 1, dict()),                                           # reference default preset, textures
    (0.9, 0, dict(time=3.7)),                                   # animated: phase rotation, twinkle, jets
    (-0.7, 1, dict(features=7 | 64, show_redshift=1.0)),        # redshift overlay, retrograde spin
    (0.9, 0, dict(quality=0)),                                  # RAY_QUALITY_LOW indicator path
    (0.9, 1, dict(cam_pos=(0.0, 6.0, -60.0), cam_quat=(0.05, 0.0, 0.0, 0.99875))),  # SAB camera
    (0.9, 0, dict(features=6, turbulence=0.75)),                # lensing off: straight rays through the disk
    (0.999, 0, dict(features=1)),                               # lensing only (the march without any sampling)
    (0.9, 1, dict(lensing_strength=0.5, time=1.3)),             # u_lensing_strength != 1 (folded scalings of FAST)
])
@pytest.mark.parametrize("arith", [0, 1])  # shader operation order / FAST contract
def test_glsl_kernel_matches_oracle(engine_mod, oracle, spin, tone, kw, arith):
    import torch
    W, H = 480, 270
    gp = engine_mod.glsl_params(W, H, 1.0, spin, max_ray_steps=512, tone_map=tone, arith=arith, **kw)
    n = W * H
    with engine_mod.PhysicsEngine(1.0, spin) as e:
        rgba = torch.zeros(n, 4, dtype=torch.float32, device="cuda:0")
        steps = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        tot = e.render_frame_glsl(gp, rgba, steps)
    ref_rgba, ref_steps = oracle.glsl_frame(oracle.glsl_params_from(gp), nthreads=8)
    assert tot == int(steps.sum().item())
    _compare(rgba.cpu().numpy().reshape(H, W, 4), steps.cpu().numpy().reshape(H, W), ref_rgba, ref_steps,
             exact=(arith == 0))


@pytest.mark.gpu
@pytest.mark.parametrize("mass", [12.4, 12.6, 14.0])   # r_h = 17.80 / 18.09 / 20.10 at a = 0.9 M
@pytest.mark.parametrize("arith", [0, 1])
def test_glsl_far_field_step_of_a_large_hole(engine_mod, oracle, mass, arith):
    """The FAST march replaces the shader's r > 30 block (fragment.glsl.ts:152-156) by a min with 3.0 on
    the photon-sphere limit (glsl_fragment.hpp; the identity over every f32 radius:
    tests/test_glsl_fast_identities.py).  Holes large enough that the block's compare, boost and clamp all
    see radii near r = 30, against the oracle: shader order bit for bit, FAST to the usual bars."""
    import torch
    W, H = 320, 180
    gp = engine_mod.glsl_params(W, H, mass, 0.9, max_ray_steps=512, arith=arith, zoom=300.0)
    with engine_mod.PhysicsEngine(mass, 0.9) as e:
        rgba = torch.zeros(W * H, 4, dtype=torch.float32, device="cuda:0")
        steps = torch.zeros(W * H, dtype=torch.int32, device="cuda:0")
        tot = e.render_frame_glsl(gp, rgba, steps)
    ref_rgba, ref_steps = oracle.glsl_frame(oracle.glsl_params_from(gp), nthreads=8)
    assert tot == int(steps.sum().item()) and tot > 50 * W * H   # the rays do march, from r = 300 > 30
    _compare(rgba.cpu().numpy().reshape(H, W, 4), steps.cpu().numpy().reshape(H, W), ref_rgba, ref_steps,
             exact=(arith == 0))


_ORACLE_FRAMES = {}


@pytest.mark.gpu
@pytest.mark.parametrize("arith", [0, 1])
def test_config2_bench_form_against_the_oracle(engine_mod, oracle, arith):
    """BASELINE configs[1] exactly as `bench.py --config c2` runs it -- 1920x1080, u_maxRaySteps = 512
    (the shader clamps to 500), the reference's default preset with its noise textures, jets, stars and
    glows -- EVERY pixel against the shader oracle (the C oracle marches the frame's 786 M steps in a
    few seconds on 16 cores).  arith 0 = shader order: the checker's step counts and colours bit for
    bit on all 2 073 600 pixels; arith 1 = the FAST form the bench line runs: FAST_BARS, plus the
    measured figures to GRV_C2_JSON=<path> (profiles/r04_full_frame_parity_c2.jsonl)."""
    import json
    import os
    import torch
    W, H = 1920, 1080
    gp = engine_mod.glsl_params(W, H, 1.0, 0.999, max_ray_steps=512, arith=arith)
    with engine_mod.PhysicsEngine(1.0, 0.999) as e:
        rgba = torch.zeros(W * H, 4, dtype=torch.float32, device="cuda:0")
        steps = torch.zeros(W * H, dtype=torch.int32, device="cuda:0")
        tot = e.render_frame_glsl(gp, rgba, steps)
        assert tot == int(steps.sum(dtype=torch.int64).item())
        g, s = rgba.view(H, W, 4).cpu().numpy(), steps.view(H, W).cpu().numpy()
    if "c2" not in _ORACLE_FRAMES:  # one oracle frame serves both contracts (the uniforms are the same)
        _ORACLE_FRAMES["c2"] = oracle.glsl_frame(oracle.glsl_params_from(gp), nthreads=16)
    ref_rgba, ref_steps = _ORACLE_FRAMES["c2"]
    m = _compare(g, s, ref_rgba, ref_steps, exact=(arith == 0))
    path = os.environ.get("GRV_C2_JSON")
    if path:
        rec = {"frame": "1920x1080 GLSL fragment march, default preset, u_maxRaySteps=512, a=0.999",
               "arith": ["shader order", "FAST"][arith], "pixels_compared": int(s.size), "oracle_steps": int(ref_steps.sum()),
               "engine_steps": int(tot)}
        rec.update(m if m else {"bit_identical": True})
        with open(path, "a") as f:
            f.write(json.dumps(rec) + "\n")


@pytest.mark.gpu
def test_glsl_kernel_shadow_guide_and_custom_textures(engine_mod, oracle):
    import torch
    W, H = 320, 180
    curve = oracle.bardeen_shadow(1.0, 0.9, np.deg2rad(97.0), 32)
    gp = engine_mod.glsl_params(W, H, 1.0, 0.9, max_ray_steps=300, show_kerr_shadow=1.0,
                                shadow_curve=curve[:64], tone_map=1)
    noise, blue = engine_mod.seeded_noise_rgba8(77), engine_mod.seeded_noise_rgba8(78)
    with engine_mod.PhysicsEngine(1.0, 0.9) as e:
        e.set_glsl_noise(noise, blue)
        rgba = torch.zeros(W * H, 4, dtype=torch.float32, device="cuda:0")
        steps = torch.zeros(W * H, dtype=torch.int32, device="cuda:0")
        e.render_frame_glsl(gp, rgba, steps)
    ref_rgba, ref_steps = oracle.glsl_frame(oracle.glsl_params_from(gp, noise, blue), nthreads=8)
    _compare(rgba.cpu().numpy().reshape(H, W, 4), steps.cpu().numpy().reshape(H, W), ref_rgba, ref_steps, exact=True)
    green = (ref_rgba[..., 1] > 0.9) & (ref_rgba[..., 0] < 0.2)
    assert green.any()


@pytest.mark.gpu
def test_wgsl_packed_kernel_odd_sizes_and_tiles(engine_mod):
    """Two rays per lane: odd slot counts, ragged frames and tile subsets must leave no pixel behind
    and agree with the one-ray FAST kernel to rounding (same contract, same expressions)."""
    import torch
    for W, H, world in ((97, 61, 1), (200, 130, 3), (64, 64, 1)):
        cam = engine_mod.camera_look_at(EYE, aspect=W / H)
        with engine_mod.PhysicsEngine(1.0, 0.9) as e:
            for r in range(world):
                outs = {}
                for arith in (1, 2):
                    gp = engine_mod.wgsl_params(W, H, cam, 1.0, 0.9, max_steps=200, arith=arith, stars=0,
                                                tile_world=world, tile_rank=r)
                    gp_ray_count = engine_mod.render_params(W, H, tile_world=world, tile_rank=r)
                    n = e.frame_ray_count(gp_ray_count)
                    rgba = torch.full((n, 4), -7.0, dtype=torch.float32, device="cuda:0")
                    steps = torch.zeros(n, dtype=torch.int32, device="cuda:0")
                    tot = e.render_frame_wgsl(gp, rgba, steps)
                    torch.cuda.synchronize()
                    outs[arith] = (rgba.cpu().numpy(), steps.cpu().numpy(), tot)
                a, b = outs[1], outs[2]
                assert a[2] == int(a[1].sum()) and b[2] == int(b[1].sum())
                written = a[0][:, 3] == 1.0
                assert np.array_equal(written, b[0][:, 3] == 1.0)           # the same pixels are covered
                assert (np.abs(a[1].astype(np.int64) - b[1]) <= 1).mean() >= 0.995
                assert np.abs(a[0][written] - b[0][written]).max() <= 5e-2 * max(a[0][written][:, :3].max(), 1e-9)


@pytest.mark.gpu
def test_shader_kernels_tile_partition(engine_mod):
    """Config-4 shape in miniature: tiles of two ranks reassemble the whole frame bitwise."""
    import torch
    from blackhole_simulation_amd import distributed as D
    W, H = 320, 200
    with engine_mod.PhysicsEngine(1.0, 0.999) as e:
        gp = engine_mod.glsl_params(W, H, 1.0, 0.999, max_ray_steps=300)
        whole = torch.zeros(W * H, 4, dtype=torch.float32, device="cuda:0")
        tot = e.render_frame_glsl(gp, whole)
        img = torch.zeros(H, W, 4, dtype=torch.float32, device="cuda:0")
        tsum = 0
        rp = engine_mod.render_params(W, H)
        for r in range(2):
            gpr = engine_mod.glsl_params(W, H, 1.0, 0.999, max_ray_steps=300, tile_world=2, tile_rank=r)
            nt = len(D.tiles_of_rank(W, H, 2, r)) * 4096
            buf = torch.zeros(nt, 4, dtype=torch.float32, device="cuda:0")
            tsum += e.render_frame_glsl(gpr, buf)
            e.unpack_tiles_device(D.rank_params(rp, 2, r), r, buf, img, 16)
        torch.cuda.synchronize()
    assert tsum == tot
    assert torch.equal(img.reshape(-1, 4), whole)


@pytest.mark.gpu
def test_measured_dispatch_order_never_changes_a_pixel(engine_mod):
    """The dispatched FAST marches start their blocks in the order last frame's block durations suggest
    (longest first; csrc/engine_types.hpp MarchSched, one counting sort per frame).  Any order must give the
    same image: frame 1 runs in natural order, frames 2.. in measured orders (different every time --
    durations are wall-clock), on one stream and alternating two, through a resize and back."""
    import torch
    bh = engine_mod
    with bh.PhysicsEngine(1.0, 0.999) as e:
        for kind in ("glsl", "wgsl"):
            want = {}
            for rep, (W, H) in enumerate([(640, 360), (640, 360), (333, 211), (640, 360), (640, 360), (640, 360)]):
                cam = bh.camera_look_at(EYE, aspect=W / H)
                rgba = torch.full((W * H, 4), -3.0, dtype=torch.float32, device="cuda:0")
                steps = torch.zeros(W * H, dtype=torch.int32, device="cuda:0")
                stream = torch.cuda.Stream() if rep % 2 else torch.cuda.current_stream()
                with torch.cuda.stream(stream):
                    if kind == "glsl":
                        gp = bh.glsl_params(W, H, 1.0, 0.999, max_ray_steps=300, arith=bh.ARITH_FAST)
                        tot = e.render_frame_glsl(gp, rgba, steps, stream=stream.cuda_stream)
                    else:
                        wp = bh.wgsl_params(W, H, cam, 1.0, 0.999, max_steps=300, arith=bh.ARITH_FAST_PACKED)
                        tot = e.render_frame_wgsl(wp, rgba, steps, stream=stream.cuda_stream)
                torch.cuda.synchronize()
                got = (rgba.cpu().numpy().view(np.uint32), steps.cpu().numpy(), tot)
                if (W, H) not in want:
                    want[(W, H)] = got
                    assert tot == int(got[1].sum()) and (got[0] != np.float32(-3.0).view(np.uint32)).all()
                else:
                    assert np.array_equal(got[0], want[(W, H)][0]) and np.array_equal(got[1], want[(W, H)][1]), (kind, rep)
                    assert got[2] == want[(W, H)][2]


@pytest.mark.gpu
@pytest.mark.parametrize("arith", [1, 2])
def test_fast_marches_on_the_disk_edge_knife_edge(engine_mod, oracle, arith):
    """Named edge case of the randomised FAST campaign (profiles/r05_fuzz_fast.txt).  A camera IN the
    equatorial plane at EXACTLY the disk's outer radius: theta0 = pi/2 makes the crossing test
    (theta_before - pi/2)(theta_after - pi/2) <= 0 true on the very first step of every ray, and whether that
    crossing shades depends on `r_before < 30.0` (compute.wgsl.ts:216-217) with r_before = |camera| = 30 to the
    ulp -- shader order (correctly rounded sqrt) gets 30.000002 and shades nothing, the FAST forms' |camera|
    comes out one ulp lower and every pixel of the frame starts with a disk sample.  Both are the shader's
    algorithm on an input that sits on its discontinuity: the FAST frame must equal, to FAST_BARS, the
    shader-order frame of the camera moved 3e-7 (relative) inwards -- or the one at 30 itself."""
    import torch
    W, H, budget = 213, 134, 512
    th, ph = np.pi / 2, 5.151
    frames = {}
    for name, r0 in (("at", 30.0), ("inside", 30.0 * (1.0 - 3e-7))):
        eye = (r0 * np.sin(th) * np.cos(ph), r0 * np.cos(th), r0 * np.sin(th) * np.sin(ph))
        cam = engine_mod.camera_look_at(eye, fovy_deg=25.0, aspect=W / H)
        gp = engine_mod.wgsl_params(W, H, cam, 1.0, 0.999, max_steps=budget, arith=arith, stars=0)
        frames[name] = oracle.wgsl_frame(oracle.wgsl_params_from(gp), nthreads=4)
        if name == "at":
            with engine_mod.PhysicsEngine(1.0, 0.999) as e:
                rgba = torch.zeros(W * H, 4, dtype=torch.float32, device="cuda:0")
                steps = torch.zeros(W * H, dtype=torch.int32, device="cuda:0")
                e.render_frame_wgsl(gp, rgba, steps)
                got = (rgba.cpu().numpy().reshape(H, W, 4), steps.cpu().numpy().reshape(H, W))
    # the two shader-order frames differ all over (the first-step sample is there or not) ...
    d = np.abs(frames["at"][0] - frames["inside"][0])[..., :3].max(-1)
    assert (d > 1e-3).mean() > 0.5
    # ... and the FAST frame is one of them
    errs = []
    for name in ("at", "inside"):
        try:
            _compare(got[0], got[1], frames[name][0], frames[name][1])
            return
        except AssertionError as exc:
            errs.append((name, str(exc)[:200]))
    raise AssertionError(errs)
