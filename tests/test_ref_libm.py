"""oracle/ref_libm.c: the specified sin / cos / pow of the f64 ray path (fdlibm-lineage routines,
IEEE + - * / sqrt only).  CPU: accuracy against mpmath, IEEE special cases, and pinned bit
patterns (the routines are pure functions of their argument: the same bits on any host).
GPU: the STRICT kernels' copies (csrc/strict_libm.hpp) return the same bits."""
import numpy as np
import pytest


def _ulps(got, exact, mp):
    if exact == 0:
        return abs(got)
    u = mp.mpf(2) ** (mp.floor(mp.log(abs(exact), 2)) - 52)
    return float(abs(mp.mpf(got) - exact) / u)


def _sincos_args(rng, n):
    return np.concatenate([
        rng.uniform(-8, 8, n), rng.uniform(-200, 200, n), rng.uniform(-1e5, 1e5, n // 2),
        rng.uniform(-1e-3, 1e-3, n // 4), 10.0 ** rng.uniform(-30, -3, n // 4),
        np.array([0.0, -0.0, np.pi / 4, np.pi / 2, np.pi, 3 * np.pi / 2, 2 * np.pi, 0.7853981633974483,
                  0.7853981633974484, 1.5707963267948966, 3.141592653589793, 6.283185307179586,
                  1e-300, 5e-324, 710.0, 1e6, 1.6e6])])


def _pow_args(rng, n):
    ys = np.array([-0.2, -0.25, 1.5, 0.75, 0.25, 2.5, 0.4, 1.0 / 3.0, -3.7, 7.1, 0.5, 2.0, -1.0, 3.0])
    x = np.concatenate([10.0 ** rng.uniform(-12, 8, n), rng.uniform(0.5, 2.0, n // 2),
                        10.0 ** rng.uniform(-300, 300, n // 4)])
    y = rng.choice(ys, x.size)
    return x, y


def test_accuracy_against_mpmath(oracle):
    mp = pytest.importorskip("mpmath")
    mp.mp.prec = 200
    rng = np.random.default_rng(2024)
    worst = 0.0
    for x in _sincos_args(rng, 600):
        worst = max(worst, _ulps(float(oracle.ref_sin(x)), mp.sin(mp.mpf(float(x))), mp),
                    _ulps(float(oracle.ref_cos(x)), mp.cos(mp.mpf(float(x))), mp))
    assert worst < 1.0, worst
    x, y = _pow_args(rng, 1500)
    worst_p = 0.0
    for a, b in zip(x, y):
        ex = mp.power(mp.mpf(float(a)), mp.mpf(float(b)))
        if ex > mp.mpf(2) ** 1023 or ex < mp.mpf(2) ** -1021:
            continue  # overflow / subnormal results: covered by the special-case test
        worst_p = max(worst_p, _ulps(float(oracle.ref_pow(a, b)), ex, mp))
    assert worst_p < 1.0, worst_p
    worst_e = 0.0
    for x in np.concatenate([rng.uniform(-708, 709.7, 800), rng.uniform(-2, 2, 400), [0.0, 1.0, -1.0, 0.34657359027997264]]):
        worst_e = max(worst_e, _ulps(float(oracle.ref_exp(x)), mp.exp(mp.mpf(float(x))), mp))
    assert worst_e < 1.0, worst_e
    worst_a = 0.0
    for x in np.concatenate([rng.uniform(-5, 5, 800), 10.0 ** rng.uniform(-12, 20, 300), [0.4375, 0.6875, 1.1875, 2.4375, 1.0]]):
        worst_a = max(worst_a, _ulps(float(oracle.ref_atan(x)), mp.atan(mp.mpf(float(x))), mp))
    assert worst_a < 1.0, worst_a
    worst_l = worst_c = worst_2 = 0.0
    for x in np.concatenate([10.0 ** rng.uniform(-300, 300, 500), rng.uniform(0.5, 2.0, 500), [5e-324, 2.0, 0.5]]):
        worst_l = max(worst_l, _ulps(float(oracle.ref_fn("log", x)), mp.log(mp.mpf(float(x))), mp))
    for x in np.concatenate([rng.uniform(-1, 1, 800), 1 - 10.0 ** rng.uniform(-16, -1, 100), [0.5, -0.5, 0.0]]):
        worst_c = max(worst_c, _ulps(float(oracle.ref_fn("acos", x)), mp.acos(mp.mpf(float(x))), mp))
    for y, x in zip(rng.uniform(-5, 5, 800) * 10.0 ** rng.uniform(-3, 3, 800), rng.uniform(-5, 5, 800)):
        worst_2 = max(worst_2, _ulps(float(oracle.ref_fn("atan2", y, x)), mp.atan2(mp.mpf(float(y)), mp.mpf(float(x))), mp))
    assert worst_l < 1.0 and worst_c < 1.0 and worst_2 < 1.5, (worst_l, worst_c, worst_2)
    # f32 forms = the f64 routine rounded once: within half an f32 ulp (+ the f64 error) of the truth
    for name, fn, xs in (("sinf", mp.sin, rng.uniform(-50, 50, 300)), ("expf", mp.exp, rng.uniform(-80, 80, 300)),
                         ("logf", mp.log, 10.0 ** rng.uniform(-30, 30, 300)), ("acosf", mp.acos, rng.uniform(-1, 1, 300))):
        for x in xs.astype(np.float32):
            got, ex = float(oracle.ref_fn(name, x)), fn(mp.mpf(float(x)))
            if ex != 0 and mp.mpf(2) ** -120 < abs(ex) < mp.mpf(2) ** 120:
                assert abs(mp.mpf(got) - ex) <= mp.mpf(2) ** (mp.floor(mp.log(abs(ex), 2)) - 23) * mp.mpf("0.5000001"), (name, x)


def test_ieee_special_cases(oracle):
    inf, nan = np.inf, np.nan
    with np.errstate(all="ignore"):
        for x, y in ((1e300, 2.5), (1e-300, 2.5), (2.0, -1074.0), (2.0, -1080.0), (0.0, -0.2), (-0.0, -3.0),
                     (inf, -0.2), (inf, 0.2), (-inf, 3.0), (-inf, 2.0), (-8.0, 1.0 / 3.0), (-8.0, 3.0), (-2.0, 2.0),
                     (nan, 0.0), (1.0, nan), (nan, 1.0), (2.0, nan), (2.0, 1023.5), (2.0, 1024.0),
                     (5e-324, 0.5), (0.5, inf), (0.5, -inf), (2.0, inf), (2.0, -inf), (-1.0, inf),
                     (0.0, 0.0), (0.0, 2.0), (-0.0, 3.0), (1.0000001, 1e9), (7.0, 0.5), (3.0, 2.0)):
            got, want = float(oracle.ref_pow(x, y)), float(np.float64(x) ** np.float64(y))
            if np.isnan(want):
                assert np.isnan(got), (x, y, got)
            elif np.isinf(want) or want == 0.0 or y in (0.5, 2.0, 1.0, -1.0):
                assert got == want and np.signbit(got) == np.signbit(want), (x, y, got, want)
            else:
                assert abs(got - want) <= 2e-16 * abs(want) + 5e-324, (x, y, got, want)
    for f in (oracle.ref_sin, oracle.ref_cos):
        assert np.isnan(float(f(inf))) and np.isnan(float(f(-inf))) and np.isnan(float(f(nan)))
    with np.errstate(all="ignore"):
        for x in (inf, -inf, 710.0, -746.0, -745.0, 0.0, -0.0, 1e-300, 709.782712893384, -745.1332191019411):
            assert float(oracle.ref_exp(x)) == float(np.exp(x)), x
    assert np.isnan(float(oracle.ref_exp(nan))) and np.isnan(float(oracle.ref_atan(nan)))
    for x in (inf, -inf, 1e70, -1e70, 0.0, 1e-300):
        assert float(oracle.ref_atan(x)) == float(np.arctan(x)), x
    assert np.signbit(float(oracle.ref_atan(-0.0)))
    with np.errstate(all="ignore"):
        for y, x in ((0.0, 1.0), (-0.0, 1.0), (0.0, -1.0), (-0.0, -1.0), (1.0, 0.0), (-1.0, 0.0), (inf, inf), (-inf, inf),
                     (inf, -inf), (-inf, -inf), (1.0, inf), (-1.0, inf), (1.0, -inf), (-1.0, -inf), (inf, 1.0),
                     (1e300, 1e-300), (1e-300, -1e300), (0.0, 0.0), (-0.0, -0.0), (0.0, -0.0)):
            got, want = float(oracle.ref_fn("atan2", y, x)), float(np.arctan2(y, x))
            assert got == want and np.signbit(got) == np.signbit(want), (y, x)
        for x in (0.0, -0.0, inf, 1.0, 5e-324):
            assert float(oracle.ref_fn("log", x)) == float(np.log(x)), x
        assert np.isnan(float(oracle.ref_fn("log", -1.0))) and np.isnan(float(oracle.ref_fn("acos", 1.5)))
        assert float(oracle.ref_fn("acos", 1.0)) == 0.0 and float(oracle.ref_fn("acos", -1.0)) == np.pi
    assert float(oracle.ref_sin(0.0)) == 0.0 and np.signbit(float(oracle.ref_sin(-0.0)))
    assert float(oracle.ref_cos(0.0)) == 1.0 and float(oracle.ref_sin(1e-300)) == 1e-300
    # odd / even symmetry is exact
    xs = np.random.default_rng(1).uniform(-50, 50, 200)
    assert np.array_equal(oracle.ref_sin(-xs), -oracle.ref_sin(xs))
    assert np.array_equal(oracle.ref_cos(-xs), oracle.ref_cos(xs))


# bit patterns recorded from oracle/ref_libm.c; any IEEE-754 host must reproduce them
PINS = (("sin", 1.0, None), ("cos", 1.0, None), ("sin", 100.0, None), ("cos", 1.5707963267948966, None),
        ("sin", 3.141592653589793, None), ("cos", 12345.678, None), ("pow", 0.37, -0.2),
        ("pow", 123.456, -0.25), ("pow", 9.5, 1.5), ("pow", 0.015625, 0.75), ("pow", 0.3, 0.4),
        ("exp", 1.0, None), ("exp", -37.25, None), ("exp", 700.5, None), ("exp", -730.0, None),
        ("atan", 0.3, None), ("atan", 0.6, None), ("atan", 1.0, None), ("atan", 2.0, None), ("atan", -77.7, None),
        ("log", 10.0, None), ("log", 1.0000001, None), ("log", 3e-310, None), ("acos", 0.3, None), ("acos", -0.7, None),
        ("acos", 0.99, None), ("atan2", 1.0, -2.0), ("atan2", -3.0, 0.5))
PIN_HEX = (
    '0x1.aed548f090ceep-1',
    '0x1.14a280fb5068cp-1',
    '-0x1.03425b78c4db8p-1',
    '0x1.1a62633145c07p-54',
    '0x1.1a62633145c07p-53',
    '0x1.6b94c3bbe24b8p-1',
    '0x1.3851e33fd110fp+0',
    '0x1.3333536991f84p-2',
    '0x1.d47ed6be5578ap+4',
    '0x1.6a09e667f3bccp-5',
    '0x1.3c5064a1418b7p-1',
    '0x1.5bf0a8b14576ap+1',
    '0x1.3278bcd70e981p-54',
    '0x1.8625c7d4f56c2p+1010',
    '0x0.00000001c7ea3p-1022',
    '0x1.2a73a661eaf06p-2',
    '0x1.14b1dd5f90ce1p-1',
    '0x1.921fb54442d18p-1',
    '0x1.1b6e192ebbe44p+0',
    '-0x1.8ed44e3384d23p+0',
    '0x1.26bb1bbb55516p+1',
    '0x1.ad7f2847b6492p-24',
    '-0x1.6459f44103e87p+9',
    '0x1.441f5ecbeef59p+0',
    '0x1.2c501446cd5f2p+1',
    '0x1.21df72882bfd8p-3',
    '0x1.56c6e7397f5aep+1',
    '-0x1.67d8863bc99bdp+0')


def _eval(oracle, name, x, y):
    return float(oracle.ref_fn(name, x, y))


def test_pinned_bit_patterns(oracle):
    for (name, x, y), want in zip(PINS, PIN_HEX):
        assert _eval(oracle, name, x, y).hex() == want, (name, x, y)


def _same_bits(got, want):
    ok = ~np.isnan(want)
    return (np.array_equal(got[ok], want[ok]) and np.array_equal(np.signbit(got[ok]), np.signbit(want[ok]))
            and np.array_equal(np.isnan(got), np.isnan(want)))


def test_engine_host_routines_return_the_oracle_bits(engine_mod, oracle):
    """csrc/strict_libm.hpp (product, host build: no device needed) against oracle/ref_libm.c."""
    E = engine_mod.engine
    rng = np.random.default_rng(9)
    inf, nan = np.inf, np.nan
    xs = np.concatenate([_sincos_args(rng, 4000), [inf, -inf, nan]])
    assert _same_bits(E.strict_math_host(E.MATH_SIN, xs), oracle.ref_fn("sin", xs))
    assert _same_bits(E.strict_math_host(E.MATH_COS, xs), oracle.ref_fn("cos", xs))
    px, py = _pow_args(rng, 6000)
    px = np.concatenate([px, [0.0, -0.0, inf, -inf, nan, 1.0, -1.0, -8.0, 5e-324, 1e300, 2.0, 2.0, 0.5, 1.0000001]])
    py = np.concatenate([py, [-0.2, -3.0, -0.2, 3.0, 0.0, nan, inf, 3.0, 0.3, 2.5, -1074.0, 1023.5, inf, 1e9]])
    assert _same_bits(E.strict_math_host(E.MATH_POW, px, py), oracle.ref_fn("pow", px, py))
    ex = np.concatenate([rng.uniform(-750, 712, 5000), rng.uniform(-2, 2, 2000), [0.0, -0.0, inf, -inf, nan, 1e-300]])
    assert _same_bits(E.strict_math_host(E.MATH_EXP, ex), oracle.ref_fn("exp", ex))
    lx = np.concatenate([10.0 ** rng.uniform(-320, 300, 4000), rng.uniform(0.5, 2, 2000), [0.0, -0.0, -1.0, inf, nan, 5e-324]])
    assert _same_bits(E.strict_math_host(E.MATH_LOG, lx), oracle.ref_fn("log", lx))
    ax = np.concatenate([rng.uniform(-5, 5, 4000), 10.0 ** rng.uniform(-20, 30, 1000), [0.0, -0.0, inf, -inf, nan, 0.4375, 0.6875, 1.1875, 2.4375]])
    assert _same_bits(E.strict_math_host(E.MATH_ATAN, ax), oracle.ref_fn("atan", ax))
    cx = np.concatenate([rng.uniform(-1, 1, 5000), 1 - 10.0 ** rng.uniform(-16, -1, 500), [1.0, -1.0, 0.0, 1.5, nan]])
    assert _same_bits(E.strict_math_host(E.MATH_ACOS, cx), oracle.ref_fn("acos", cx))
    ay = np.concatenate([rng.uniform(-5, 5, 5000) * 10.0 ** rng.uniform(-3, 3, 5000), [0.0, -0.0, 0.0, 1.0, inf, -inf, 1e300]])
    axx = np.concatenate([rng.uniform(-5, 5, 5000), [1.0, -1.0, -0.0, 0.0, inf, -inf, 1e-300]])
    assert _same_bits(E.strict_math_host(E.MATH_ATAN2, ay, axx), oracle.ref_fn("atan2", ay, axx))
    f = np.float32
    for op, name, a, b in ((E.MATH_SIN, "sinf", rng.uniform(-100, 100, 3000), None), (E.MATH_EXP, "expf", rng.uniform(-110, 95, 3000), None),
                           (E.MATH_LOG, "logf", 10.0 ** rng.uniform(-44, 38, 3000), None), (E.MATH_ACOS, "acosf", rng.uniform(-1, 1, 3000), None),
                           (E.MATH_POW, "powf", 10.0 ** rng.uniform(-6, 6, 3000), rng.uniform(-8, 8, 3000)),
                           (E.MATH_ATAN2, "atan2f", rng.uniform(-9, 9, 3000), rng.uniform(-9, 9, 3000))):
        a32, b32 = a.astype(f), None if b is None else b.astype(f)
        got = E.strict_math_host(op | E.MATH_F32, a32.astype(np.float64), None if b is None else b32.astype(np.float64))
        assert _same_bits(got.astype(f), oracle.ref_fn(name, a32, b32)), name


@pytest.mark.gpu
def test_device_routines_return_the_same_bits(engine_mod, oracle):
    bh = engine_mod
    rng = np.random.default_rng(77)
    xs = _sincos_args(rng, 40000)
    xs = np.concatenate([xs, [np.inf, -np.inf, np.nan]])
    px, py = _pow_args(rng, 60000)
    sx = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, -8.0, 5e-324, 1e300, 2.0, 2.0, 0.5, 1.0000001])
    sy = np.array([-0.2, -3.0, -0.2, 3.0, 0.0, np.nan, np.inf, 3.0, 0.3, 2.5, -1074.0, 1023.5, np.inf, 1e9])
    px, py = np.concatenate([px, sx]), np.concatenate([py, sy])
    with bh.PhysicsEngine(1.0, 0.5) as e:
        for op, ref in ((bh.engine.MATH_SIN, oracle.ref_sin), (bh.engine.MATH_COS, oracle.ref_cos),
                        (bh.engine.MATH_SINCOS_SIN, oracle.ref_sin), (bh.engine.MATH_SINCOS_COS, oracle.ref_cos)):
            got, want = e.strict_math(op, xs), ref(xs)
            assert np.array_equal(got.view(np.uint64)[~np.isnan(want)], want.view(np.uint64)[~np.isnan(want)]), op
            assert np.array_equal(np.isnan(got), np.isnan(want))
        ex = np.concatenate([rng.uniform(-750, 712, 40000), rng.uniform(-2, 2, 20000),
                             [0.0, -0.0, np.inf, -np.inf, np.nan, 1e-300, 709.782712893384, -745.1332191019411]])
        got, want = e.strict_math(bh.engine.MATH_EXP, ex), oracle.ref_exp(ex)
        assert np.array_equal(got.view(np.uint64)[~np.isnan(want)], want.view(np.uint64)[~np.isnan(want)])
        assert np.array_equal(np.isnan(got), np.isnan(want))
        ax = np.concatenate([rng.uniform(-5, 5, 40000), 10.0 ** rng.uniform(-20, 30, 10000) * rng.choice([-1.0, 1.0], 10000),
                             [0.0, -0.0, np.inf, -np.inf, np.nan, 0.4375, 0.6875, 1.1875, 2.4375]])
        got, want = e.strict_math(bh.engine.MATH_ATAN, ax), oracle.ref_atan(ax)
        assert np.array_equal(got.view(np.uint64)[~np.isnan(want)], want.view(np.uint64)[~np.isnan(want)])
        assert np.array_equal(np.isnan(got), np.isnan(want))
        E = bh.engine

        def same(got, want):
            ok = ~np.isnan(want)
            return np.array_equal(got[ok], want[ok]) and np.array_equal(np.signbit(got[ok]), np.signbit(want[ok])) \
                and np.array_equal(np.isnan(got), np.isnan(want))
        lx = np.concatenate([10.0 ** rng.uniform(-320, 300, 30000), rng.uniform(0.5, 2, 20000), [0.0, -0.0, -1.0, np.inf, np.nan, 5e-324]])
        assert same(e.strict_math(E.MATH_LOG, lx), oracle.ref_fn("log", lx))
        cx = np.concatenate([rng.uniform(-1, 1, 40000), 1 - 10.0 ** rng.uniform(-16, -1, 5000), [1.0, -1.0, 0.0, 1.5, np.nan]])
        assert same(e.strict_math(E.MATH_ACOS, cx), oracle.ref_fn("acos", cx))
        ay = np.concatenate([rng.uniform(-5, 5, 40000) * 10.0 ** rng.uniform(-3, 3, 40000), [0.0, -0.0, 0.0, 1.0, np.inf, -np.inf, 1e300]])
        axx = np.concatenate([rng.uniform(-5, 5, 40000), [1.0, -1.0, -0.0, 0.0, np.inf, -np.inf, 1e-300]])
        assert same(e.strict_math(E.MATH_ATAN2, ay, axx), oracle.ref_fn("atan2", ay, axx))
        # f32 forms of the shader-order kernels
        f = np.float32
        for op, name, a, b in ((E.MATH_SIN, "sinf", rng.uniform(-100, 100, 30000), None),
                               (E.MATH_COS, "cosf", rng.uniform(-100, 100, 30000), None),
                               (E.MATH_EXP, "expf", rng.uniform(-110, 95, 30000), None),
                               (E.MATH_LOG, "logf", 10.0 ** rng.uniform(-44, 38, 30000), None),
                               (E.MATH_ACOS, "acosf", rng.uniform(-1, 1, 30000), None),
                               (E.MATH_POW, "powf", 10.0 ** rng.uniform(-6, 6, 30000), rng.uniform(-8, 8, 30000)),
                               (E.MATH_ATAN2, "atan2f", rng.uniform(-9, 9, 30000), rng.uniform(-9, 9, 30000))):
            a32 = a.astype(f)
            b32 = None if b is None else b.astype(f)
            got = e.strict_math(op | E.MATH_F32, a32.astype(np.float64), None if b is None else b32.astype(np.float64))
            assert same(got.astype(f), oracle.ref_fn(name, a32, b32)), name
        got, want = e.strict_math(bh.engine.MATH_POW, px, py), oracle.ref_pow(px, py)
        ok = ~np.isnan(want)
        assert np.array_equal(got.view(np.uint64)[ok], want.view(np.uint64)[ok])
        assert np.array_equal(np.isnan(got), np.isnan(want))


@pytest.mark.gpu
def test_shared_reciprocal_quotient_is_the_ieee_quotient_inside_its_range(engine_mod):
    """SharedDiv (csrc/kerr_device.hpp): the STRICT Kerr-Schild kernels evaluate the reciprocal
    refinement of the compiler's f64 division once per denominator and share it between the quotients
    over it.  The kernels admit a right-hand side to that form only when every operand is zero or
    moderate in magnitude (divs_ok_hole / divs_ok_point: denominators in [2^-266, 2^82], numerators zero
    or in [2^-290, 2^83]); inside that range the form must return the bits of the device's own `/`
    and of the host's IEEE quotient.  4 M random operand pairs over the whole admitted range, the
    range's corners, signed zeros."""
    bh = engine_mod
    E = bh.engine
    rng = np.random.default_rng(20260930)
    n = 4_000_000

    def operands(k, lo, hi, zero_frac):
        mant = 1.0 + rng.random(k)
        mant[rng.random(k) < 0.02] = 1.0                       # exact powers of two
        mant[rng.random(k) < 0.02] = np.nextafter(2.0, 1.0)    # all-ones mantissa
        x = np.ldexp(mant, rng.integers(lo, hi, k)) * rng.choice([-1.0, 1.0], k)
        x[rng.random(k) < zero_frac] = 0.0
        return x
    num = operands(n, -290, 83, 0.03)
    den = operands(n, -266, 82, 0.0)
    corners = np.array([(a, b) for a in (2.0 ** -290, np.nextafter(2.0 ** 83, 0), 0.0, -0.0, 1.0, 3.0)
                        for b in (2.0 ** -266, -np.nextafter(2.0 ** 82, 0), 1.0, 3.0, 2.0 ** -54)])
    num = np.concatenate([num, corners[:, 0]])
    den = np.concatenate([den, corners[:, 1]])
    with bh.PhysicsEngine(1.0, 0.9) as e:
        ieee = e.strict_math(E.MATH_DIV, num, den)
        shared = e.strict_math(E.MATH_DIV_SHARED, num, den)
    host = E.strict_math_host(E.MATH_DIV, num, den)
    assert np.array_equal(host.view(np.uint64), (num / den).view(np.uint64))
    assert np.array_equal(ieee.view(np.uint64), host.view(np.uint64))
    bad = np.flatnonzero(shared.view(np.uint64) != ieee.view(np.uint64))
    assert bad.size == 0, (bad.size, num[bad[:4]], den[bad[:4]], shared[bad[:4]], ieee[bad[:4]])
    # outside the admitted range the two forms need not agree (subnormal quotients lose the scaling the
    # full sequence applies): that is what the kernels' range test is for.  Only finiteness of the
    # statement is checked here -- a zero denominator or non-finite operand maps like the full sequence
    sp_n = np.array([1.0, -1.0, 0.0, np.inf, np.nan, 1.0])
    sp_d = np.array([0.0, 0.0, 0.0, 2.0, 1.0, np.inf])
    with bh.PhysicsEngine(1.0, 0.9) as e:
        a, b = e.strict_math(E.MATH_DIV, sp_n, sp_d), e.strict_math(E.MATH_DIV_SHARED, sp_n, sp_d)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])


@pytest.mark.gpu
def test_no_fixup_quotient_and_literal_reciprocal_forms(engine_mod):
    """Round 3 forms of the STRICT divisions (csrc/kerr_device.hpp).
    SharedDivNoFixup: SharedDiv without v_div_fixup, admitted by divs_nf only for positive moderate
    denominators and numerators that are +0 or non-zero moderate -- there it must return the IEEE
    quotient.  ConstDen: the Fehlberg stage scales c h / D with D's refined reciprocal as a literal; the
    literal must be the value the device's own refinement yields (RCP_R2), and the quotient the IEEE one
    for every numerator const_div_ok admits (zero, non-finite, 2^-100 <= |h| <= 2^100 times |c| <= 7296)."""
    bh = engine_mod
    E = bh.engine
    rng = np.random.default_rng(20261001)
    n = 4_000_000

    def moderate(k, lo, hi):
        mant = 1.0 + rng.random(k)
        mant[rng.random(k) < 0.02] = 1.0
        mant[rng.random(k) < 0.02] = np.nextafter(2.0, 1.0)
        return np.ldexp(mant, rng.integers(lo, hi, k))
    num = moderate(n, -290, 83) * rng.choice([-1.0, 1.0], n)
    num[rng.random(n) < 0.03] = 0.0  # +0 only: the kernels never hand this form a -0
    den = moderate(n, -266, 82)
    dens = np.array([2197.0, 216.0, 513.0, 4104.0, 27.0, 2565.0, 40.0])
    coef = np.array([1932.0, -7200.0, 7296.0, 439.0, 3680.0, -845.0, -8.0, -3544.0, 1859.0, -11.0])
    h = moderate(n, -100, 100) * rng.choice([-1.0, 1.0], n)
    h[:8] = [0.0, -0.0, np.inf, -np.inf, np.nan, 2.0 ** -100, 2.0 ** 100, -2.0 ** 100]
    h[8:2008] = rng.uniform(1e-5, 10.0, 2000)  # the controller's range
    cn = coef[rng.integers(0, coef.size, n)] * h
    cd = dens[rng.integers(0, dens.size, n)]
    with bh.PhysicsEngine(1.0, 0.9) as e:
        nofix = e.strict_math(E.MATH_DIV_NOFIX, num, den)
        r2 = e.strict_math(E.MATH_RCP_R2, dens)
        cq = e.strict_math(E.MATH_DIV_CONST, cn, cd)
        unknown = e.strict_math(E.MATH_DIV_CONST, np.ones(2), np.array([7.0, 2196.0]))
    want = num / den
    bad = np.flatnonzero(nofix.view(np.uint64) != want.view(np.uint64))
    assert bad.size == 0, (bad.size, num[bad[:4]], den[bad[:4]], nofix[bad[:4]], want[bad[:4]])
    assert np.array_equal(r2.view(np.uint64), (1.0 / dens).view(np.uint64)), (r2, 1.0 / dens)
    with np.errstate(invalid="ignore"):
        cwant = cn / cd
    ok = ~np.isnan(cwant)
    assert np.array_equal(np.isnan(cq), np.isnan(cwant))
    bad = np.flatnonzero(cq.view(np.uint64)[ok] != cwant.view(np.uint64)[ok])
    assert bad.size == 0, (bad.size, cn[ok][bad[:4]], cd[ok][bad[:4]])
    assert np.isnan(unknown).all()
    assert np.array_equal(E.strict_math_host(E.MATH_DIV_CONST, cn[ok], cd[ok]).view(np.uint64),
                          cwant.view(np.uint64)[ok])
