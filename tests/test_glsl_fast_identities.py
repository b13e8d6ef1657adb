"""CPU checks of the two rewrites of the FAST GLSL march (csrc/glsl_fragment.hpp) that are stated as
identities in f32 arithmetic, not as tolerances -- no GPU, no oracle: numpy float32 with the kernel's
fma steps formed exactly in float64 (the product of two f32 has 48 bits; each sum below stays under 53).

1. The shader's far-field block (fragment.glsl.ts:152-156)
       if (r > 30) { dt = max(dt, 0.01 + (r - 30) * 0.08); dt = min(dt, 1.2 * 2.5); }
   after dt = clamp((r - rh) * 0.1 * distFactor, 0.01, 1.2 * distFactor), distFactor = 1 + r * 0.05,
   equals min(dt, 3.0f) for EVERY f32 radius the march can hold (1.15 r_h <= r <= 10 000), whatever
   the hole: the far boost never binds there, and above r = 30 the upper clamp is >= 3.0f.
2. The two-term twist polynomials used when the whole wave turns by less than 1/16 rad stay within
   0.51 ulp of sin and 1.01 ulp of cos (no worse than the full FAST polynomials) and within one ulp of them.
3. The far-field path: beyond r_far = max(64, r_ph + 21) and |y| >= 0.2 the whole step-size head is the
   constant 3.0f, and below 2^-12 rad the twist polynomials return exactly (angle, 1.0f).
"""
import numpy as np
import pytest

F = np.float32


def _fma(a, b, c):
    """round_f32(a * b + c) for f32 arrays, exact: products and these sums fit a float64."""
    return (a.astype(np.float64) * np.float64(b) + np.float64(c)).astype(F)


def _all_f32(lo, hi):
    """every f32 in [lo, hi] (positive), in chunks"""
    a, b = int(np.array(lo, F).view(np.uint32)), int(np.array(hi, F).view(np.uint32))
    step = 1 << 22
    for s in range(a, b + 1, step):
        yield np.arange(s, min(s + step, b + 1), dtype=np.uint32).view(F)


def _shader_dt(r, rh):
    """the FAST kernel's arithmetic for the shader's step-size head, block included (as written)"""
    dist = _fma(r, F(0.05), F(1.0))
    t1 = ((r - F(rh)) * F(0.1)) * dist
    hi = F(1.2) * dist
    dt0 = np.minimum(np.maximum(t1, F(0.01)), hi)
    far = _fma(r - F(30.0), F(0.08), F(0.01))
    dt_far = np.minimum(np.maximum(dt0, far), F(1.2) * F(2.5))
    return dt0, np.where(r > F(30.0), dt_far, dt0)


@pytest.mark.parametrize("rh", [1.0447, 2.0, 9.0, 17.9, 25.0, 40.0, 87.0, 140.0, 2000.0])
def test_far_field_block_is_a_min_with_three(rh):
    """every f32 radius between the horizon exit (r < 1.15 r_h leaves at the loop top, before the step
    size is formed) and the far exit (r > 10 000), for holes from the bench's to absurd ones"""
    assert F(1.2) * F(2.5) == F(3.0)   # the exact tie, rounded to even
    n, worst = 0, 0.0
    for r in _all_f32(max(1.0, 1.15 * rh * (1.0 - 1e-6)), 10000.0):
        dt0, dt_shader = _shader_dt(r, rh)
        assert np.array_equal(np.minimum(dt0, F(3.0)), dt_shader)
        worst = max(worst, float(dt_shader.max()))
        n += r.size
    assert worst <= 3.0 and n > 10_000_000


def test_far_boost_never_binds_inside_the_march_domain():
    """why: for r > 30 and r >= 1.15 r_h the un-clamped step 0.1 (r - r_h)(1 + 0.05 r) exceeds the far
    boost 0.01 + 0.08 (r - 30) by at least 0.67 (a quadratic in r without real roots), so max(dt, boost)
    is dt unless dt sits at its upper clamp 1.2 (1 + 0.05 r) >= 3 -- where the min with 3 decides either
    way.  The margin on a grid of (r, r_h), in float64:"""
    r = np.linspace(30.0, 10000.0, 200001)
    rh = r / 1.15                                   # the largest hole that still marches at r
    t1 = 0.1 * (r - rh) * (1.0 + 0.05 * r)
    far = 0.01 + 0.08 * (r - 30.0)
    assert (t1 - far).min() > 0.66
    # a smaller hole only makes t1 larger
    assert ((0.1 * (r - 0.5 * rh) * (1.0 + 0.05 * r)) >= t1).all()


def test_tiny_twist_polynomials():
    mp = pytest.importorskip("mpmath")
    rng = np.random.default_rng(7)
    ang = np.concatenate([rng.uniform(-0.0625, 0.0625, 3000), 0.0625 * np.array([1.0, -1.0]),
                          10.0 ** rng.uniform(-9, -2, 500)]).astype(F)
    ang = ang[np.abs(ang) <= F(0.0625)]

    z = ang * ang
    # two-term forms (glsl_fast_sincos, |ang| <= 1/16)
    s2 = _fma(ang * z, _fma(z, F(8.3333333333e-3), F(-1.6666666667e-1)), ang)
    c2 = _fma(z * z, F(4.1666666667e-2), _fma(z, F(-0.5), F(1.0)))
    # full FAST polynomials (the |ang| <= pi/4 path)
    ps = _fma(z, _fma(z, F(-1.9515295891e-4), F(8.3321608736e-3)), F(-1.6666654611e-1))
    sf = _fma(ang * z, ps, ang)
    pc = _fma(z, _fma(z, F(2.443315711809948e-5), F(-1.388731625493765e-3)), F(4.166664568298827e-2))
    cf = _fma(z * z, pc, _fma(z, F(-0.5), F(1.0)))
    mp.mp.prec = 80
    ts = np.array([float(mp.sin(mp.mpf(float(a)))) for a in ang])
    tc = np.array([float(mp.cos(mp.mpf(float(a)))) for a in ang])
    ulp_s = np.spacing(np.abs(ts).astype(F)).astype(np.float64)
    ulp_c = np.spacing(tc.astype(F)).astype(np.float64)
    assert (np.abs(s2.astype(np.float64) - ts) <= 0.51 * ulp_s).all()
    # cos: 1 - z/2 is rounded once before the z^2 term joins it -- two roundings next to 1.0, in the full
    # polynomial just the same
    assert (np.abs(c2.astype(np.float64) - tc) <= 1.01 * ulp_c).all()
    assert np.abs(c2.astype(np.float64) - tc).max() <= np.abs(cf.astype(np.float64) - tc).max() + 1e-12
    assert (np.abs(s2.astype(np.float64) - sf.astype(np.float64)) <= ulp_s).all()
    assert (np.abs(c2.astype(np.float64) - cf.astype(np.float64)) <= ulp_c).all()
    # and they agree bit for bit on nearly every angle
    assert (s2 == sf).mean() > 0.95 and (c2 == cf).mean() > 0.95


def _lim(r, rph):
    return np.minimum(_fma(np.abs(r - F(rph)), F(0.15), F(0.01)), F(3.0))


@pytest.mark.parametrize("rh,rph", [(1.0447, 1.0745), (2.0, 3.0), (1.0447, 3.9), (17.9, 50.0), (140.0, 500.0)])
def test_far_field_step_is_three(rh, rph):
    """every f32 radius from r_far to the far exit: clamp, far block, photon-sphere limit and plane
    refinement (|y| >= 0.2) together give exactly 3.0f -- what the wave-uniform far path assigns"""
    r_far = max(64.0, float(F(rph) + F(21.0)))
    n = 0
    for r in _all_f32(np.nextafter(F(max(r_far, 1.15 * rh)), F(np.inf)), 10000.0):
        _, dt = _shader_dt(r, rh)
        dt = np.minimum(dt, _lim(r, rph))
        assert (dt == F(3.0)).all()
        n += r.size
    assert n > 1_000_000
    # the plane refinement: smoothstep(0.2, 0, |y|) = 0 and the factor 1 - 0 * 0.7 = 1 for |y| >= 0.2
    y = np.concatenate([np.array([0.2, 0.20000002, 0.25, 1.0, 1e4], F), np.nextafter(F(0.2), F(1.0), dtype=F)[None]])
    t = np.clip((np.abs(y) - F(0.2)) * F(1.0 / (0.0 - 0.2)), F(0.0), F(1.0))
    h = t * t * _fma(t, F(-2.0), F(3.0))
    assert (h == 0).all() and (F(3.0) * _fma(h, F(-0.7), F(1.0)) == F(3.0)).all()


def test_twist_below_two_to_the_minus_twelve_is_the_identity_rotation():
    """|angle| < 2^-12: both polynomial tiers of glsl_fast_sincos return exactly (angle, 1.0f), which is
    what the first tier assigns.  Every f32 of the top four binades below 2^-12, and a sample below."""
    lo, hi = F(2.0 ** -16), np.nextafter(F(2.0 ** -12), F(0.0))
    for chunk in list(_all_f32(lo, hi)) + [(10.0 ** np.random.default_rng(3).uniform(-30, -4.9, 100000)).astype(F)]:
        for ang in (chunk, -chunk):
            z = ang * ang
            s2 = _fma(ang * z, _fma(z, F(8.3333333333e-3), F(-1.6666666667e-1)), ang)
            c2 = _fma(z * z, F(4.1666666667e-2), _fma(z, F(-0.5), F(1.0)))
            ps = _fma(z, _fma(z, F(-1.9515295891e-4), F(8.3321608736e-3)), F(-1.6666654611e-1))
            sf = _fma(ang * z, ps, ang)
            pc = _fma(z, _fma(z, F(2.443315711809948e-5), F(-1.388731625493765e-3)), F(4.166664568298827e-2))
            cf = _fma(z * z, pc, _fma(z, F(-0.5), F(1.0)))
            assert np.array_equal(s2, ang) and np.array_equal(sf, ang)
            assert (c2 == F(1.0)).all() and (cf == F(1.0)).all()
