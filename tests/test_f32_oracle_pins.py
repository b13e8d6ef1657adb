"""Pins of the f32 oracles (shader_oracle.c: rows a16-a18, 8f-3; post_oracle.c: row 8f-4) that do
NOT go through the HIP twin of the same transcription.  Each building block of the two marches is
called alone through a test hook and held to something it was not written from:

  * the pinned f64 oracle -- compute.wgsl.ts:42-120 states the same formulas as kerr.rs:412-499,
    which gravitas_oracle.c implements and tests/test_oracle_pins.py pins to the reference's tests;
  * the reference's own tests of the shader expressions
    (src/__tests__/physics/advanced-physics.test.ts:35-75, 77-122, 124-167, 169-195, 196-240);
  * closed forms (Schwarzschild photon acceleration, the 3 sqrt(3) M critical impact parameter);
  * the literal constants of ataa.wgsl.ts:29-86, reprojection.glsl.ts:76-113, bloom.glsl.ts:35-127
    and bloom.ts:23-39 on hand-computed 3x3 / constant images (numpy f64, written from the shader
    text).

Tolerances are f32 rounding of the expressions involved and are stated at each check.
"""
import ctypes as C

import numpy as np
import pytest


def P(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def L(oracle):
    return oracle.lib()


# --------------------------------------------------------------------------------------------
# a18: WGSL get_derivatives / symplectic_step / horizon / ISCO against the pinned f64 oracle
# --------------------------------------------------------------------------------------------
def _f64_derivative(oracle, L, m, x, p):
    s = oracle.make_state(list(x.astype(np.float64)) + list(p.astype(np.float64)))
    d = L.orc_state_derivative(C.byref(s), C.byref(m))
    return np.array(list(d.x) + list(d.p))


def test_wgsl_get_derivatives_equal_the_pinned_f64_right_hand_side(oracle, L):
    """10^4 random states (the f32 values fed to both sides): every component of
    get_derivatives (compute.wgsl.ts:42-120) within 1e-4 relative (floor 1e-3 absolute) of
    get_state_derivative over Kerr::{contravariant,hamiltonian_derivs}_ks; observed worst 2.1e-5
    -- f32 rounding through the cancelling sums of dH/dr, dH/dtheta."""
    rng = np.random.default_rng(20260930)
    worst = 0.0
    for spin in (0.0, 0.5, 0.9, 0.999):
        for M in (1.0, 2.0):
            m = oracle.metric(oracle.KERR_KS, M, spin)
            rh = L.orc_event_horizon(C.byref(m))
            for _ in range(1250):
                x = np.array([0, rng.uniform(1.05 * rh, 60), rng.uniform(0.05, np.pi - 0.05),
                              rng.uniform(-3, 3)], np.float32)
                p = np.array([-1, rng.uniform(-1.5, 1.5), rng.uniform(-6, 6), rng.uniform(-6, 6)], np.float32)
                dx, dp = np.zeros(4, np.float32), np.zeros(4, np.float32)
                L.orc_hook_wgsl_derivs(P(x), P(p), M, spin, P(dx), P(dp))
                ref = _f64_derivative(oracle, L, m, x, p)
                got = np.concatenate([dx, dp]).astype(np.float64)
                assert got[4] == 0.0 and got[7] == 0.0          # dp_t = dp_phi = 0 (hamiltonian.rs:33)
                err = np.abs(got - ref) / np.maximum(1e-3, np.abs(ref))
                worst = max(worst, err.max())
                assert err.max() <= 1e-4, (spin, M, x, p, got, ref)
    print("wgsl get_derivatives vs f64: worst relative", worst)


def test_wgsl_symplectic_step_equals_the_pinned_f64_step(oracle, L):
    """compute.wgsl.ts:122-133 is integrator.rs:209-226 (implicit midpoint, two sweeps) in f32:
    one step from 2000 random states lands within 2e-5 (relative, floor 1) of orc_step_symplectic."""
    rng = np.random.default_rng(7)
    worst = 0.0
    for spin, M in ((0.0, 1.0), (0.9, 1.0), (0.999, 1.0), (0.5, 2.0)):
        m = oracle.metric(oracle.KERR_KS, M, spin)
        rh = L.orc_event_horizon(C.byref(m))
        for _ in range(500):
            x = np.array([0, rng.uniform(1.3 * rh, 60), rng.uniform(0.1, np.pi - 0.1), rng.uniform(-3, 3)], np.float32)
            p = np.array([-1, rng.uniform(-1, 1), rng.uniform(-5, 5), rng.uniform(-5, 5)], np.float32)
            h = np.float32(np.clip((x[1] - rh) * 0.15, 0.05, 1.0))    # the kernel's own step rule
            ox, op = np.zeros(4, np.float32), np.zeros(4, np.float32)
            L.orc_hook_wgsl_step(P(x), P(p), h, M, spin, P(ox), P(op))
            s = oracle.make_state(list(x.astype(np.float64)) + list(p.astype(np.float64)))
            L.orc_step_symplectic(C.byref(s), C.byref(m), float(h))
            ref = np.array(list(s.x) + list(s.p))
            got = np.concatenate([ox, op]).astype(np.float64)
            err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
            worst = max(worst, err.max())
            assert err.max() <= 2e-5, (spin, x, p, got, ref)
    print("wgsl symplectic step vs f64: worst", worst)


def test_shader_horizon_isco_photon_sphere_equal_the_pinned_closed_forms(oracle, L):
    """compute.wgsl.ts:28-40 and chunks/metric.ts:13-37 against metric/mod.rs:75-84 and
    kerr.rs:91-123 (pinned by kerr.rs:507-554): 2e-6 relative (f32 cube roots), for the spins the
    shaders do not clamp (|a*| <= 0.999 / 0.9999)."""
    for M in (1.0, 2.5):
        for a_star in (0.0, 0.1, 0.3, 0.5, 0.7, 0.9, 0.99, 0.998):
            m = oracle.metric(oracle.KERR_KS, M, a_star)
            rh = L.orc_event_horizon(C.byref(m))
            pro, retro = L.orc_isco(C.byref(m), 0), L.orc_isco(C.byref(m), 1)
            rph = L.orc_photon_sphere(C.byref(m))
            a = a_star * M
            assert abs(L.orc_hook_wgsl_horizon(M, a) - rh) <= 2e-6 * rh
            assert abs(L.orc_hook_glsl_horizon(M, a) - rh) <= 2e-6 * rh
            assert abs(L.orc_hook_wgsl_isco(M, a) - pro) <= 2e-5 * pro       # the kernel is prograde-only
            assert abs(L.orc_hook_glsl_isco(M, a) - pro) <= 2e-5 * pro
            if a_star > 0:
                # the fragment shader separates pro- and retrograde orbits by the sign of a
                # (metric.ts:24-31) -- the polynomial mirror in advanced-physics.test.ts:23-29
                # (symmetric in a) describes an older shader
                assert abs(L.orc_hook_glsl_isco(M, -a) - retro) <= 2e-5 * retro
            assert abs(L.orc_hook_glsl_photon_sphere(M, a) - rph) <= 5e-6 * rph
    # kerr.rs:507-554 on the shader forms directly
    assert abs(L.orc_hook_glsl_isco(1.0, 0.0) - 6.0) < 1e-5 and abs(L.orc_hook_wgsl_isco(1.0, 0.0) - 6.0) < 1e-5
    assert L.orc_hook_glsl_isco(1.0, 0.998) < 1.5
    assert abs(L.orc_hook_glsl_horizon(1.0, 0.0) - 2.0) < 1e-6 and abs(L.orc_hook_glsl_horizon(1.0, 1.0) - 1.0) < 1e-6
    assert abs(L.orc_hook_glsl_photon_sphere(1.0, 0.0) - 3.0) < 1e-5


def test_wgsl_march_captures_below_and_releases_above_the_critical_impact_parameter(oracle, L):
    """Oracle-independent physics pin of the whole a18 march: equatorial rays from r0 = 60 M in
    Schwarzschild, stepped with the kernel's own step and step rule, fall in for b < 3 sqrt(3) M
    and escape for b > 3 sqrt(3) M.  Band: +-1.5 % (the coarse h near the photon sphere)."""
    M, spin = 1.0, 0.0
    bc = 3.0 * np.sqrt(3.0) * M
    m = oracle.metric(oracle.KERR_KS, M, spin)
    rh = L.orc_hook_wgsl_horizon(M, 0.0)
    outcome = {}
    for b in (0.5 * bc, 0.9 * bc, 0.985 * bc, 1.015 * bc, 1.1 * bc, 2.0 * bc):
        s = oracle.make_state([0.0, 60.0, np.pi / 2, 0.0, -1.0, -1.0, 0.0, b])
        L.orc_renormalize_null(C.byref(s), C.byref(m))        # ingoing null p_r (pinned f64 routine)
        x = np.array(list(s.x), np.float32)
        p = np.array(list(s.p), np.float32)
        end = "max"
        for _ in range(4000):
            r = float(x[1])
            if r < rh * 1.001:
                end = "horizon"
                break
            if r > 100.0:
                end = "escape"
                break
            h = np.float32(np.clip((r - rh) * 0.15, 0.05, 1.0))
            ox, op = np.zeros(4, np.float32), np.zeros(4, np.float32)
            L.orc_hook_wgsl_step(P(x), P(p), h, M, spin, P(ox), P(op))
            x, p = ox, op
        outcome[round(b / bc, 3)] = end
    assert outcome == {0.5: "horizon", 0.9: "horizon", 0.985: "horizon", 1.015: "escape", 1.1: "escape",
                       2.0: "escape"}, outcome


# --------------------------------------------------------------------------------------------
# a16 / a17: GLSL kerr_geodesic_accel, disk Doppler block, blackbody, the expressions the
# reference's own tests mirror
# --------------------------------------------------------------------------------------------
def _accel(L, p, v, M, a):
    acc, om = np.zeros(3, np.float32), C.c_float(0)
    L.orc_hook_glsl_accel(P(np.asarray(p, np.float32)), P(np.asarray(v, np.float32)), M, a, P(acc), C.byref(om))
    return acc.astype(np.float64), float(om.value)


def test_glsl_accel_at_zero_spin_is_its_closed_form(L):
    """a = 0: kerr_geodesic_accel (chunks/metric.ts:96-149) must be -p/|p| (M/r^2 + 3 M L^2/r^4),
    L = |p x v| (the second term is the Schwarzschild photon equation u'' + u = 3 M u^2,
    the first the shader's Newtonian addition) and no frame dragging.  5000 random (p, v): 2e-5
    relative of |acc|."""
    rng = np.random.default_rng(3)
    worst = 0.0
    for _ in range(5000):
        M = float(rng.choice([1.0, 0.5, 3.0]))
        p = rng.normal(size=3) * rng.uniform(2.5, 40)
        v = rng.normal(size=3)
        v /= np.linalg.norm(v)
        p32, v32 = p.astype(np.float32).astype(np.float64), v.astype(np.float32).astype(np.float64)
        got, om = _accel(L, p32, v32, M, 0.0)
        r = np.linalg.norm(p32)
        Lv = np.cross(p32, v32)
        want = -(p32 / r) * (M / r ** 2 + 3.0 * M * Lv.dot(Lv) / r ** 4)
        err = np.linalg.norm(got - want) / np.linalg.norm(want)
        worst = max(worst, err)
        assert err <= 2e-5 and om == 0.0, (p, v, got, want, om)
    print("glsl accel a=0 vs closed form: worst", worst)


def test_glsl_accel_with_spin_follows_its_stated_formula(L):
    """a != 0: radial part scaled by r_k^2 / Sigma with the oblate radius r_k of
    r^4 - (rho^2 - a^2) r^2 - a^2 y^2 = 0, L^2_eff = (L_y - a)^2 + L_x^2 + L_z^2, plus the drag
    term (y_hat x v) 2 M a / (r_k^3 + a^2 r_k) = omega, evaluated in numpy f64: 1e-4 of |acc|."""
    rng = np.random.default_rng(4)
    for _ in range(3000):
        M, a = 1.0, float(rng.uniform(-0.999, 0.999))
        p = rng.normal(size=3) * rng.uniform(2.5, 30)
        v = rng.normal(size=3)
        v /= np.linalg.norm(v)
        p, v = p.astype(np.float32).astype(np.float64), v.astype(np.float32).astype(np.float64)
        got, om = _accel(L, p, v, M, a)
        rho2, y2 = p.dot(p), p[1] ** 2
        r2 = 0.5 * ((rho2 - a * a) + np.sqrt((rho2 - a * a) ** 2 + 4 * a * a * y2))
        rk = np.sqrt(r2)
        sigma = r2 + a * a * y2 / r2
        Lv = np.cross(p, v)
        l2 = (Lv[1] - a) ** 2 + Lv[0] ** 2 + Lv[2] ** 2
        radial = -(p / np.sqrt(rho2)) * (M / r2 + 3 * M * l2 / r2 ** 2) * (r2 / sigma)
        omega = 2 * M * a / (rk ** 3 + a * a * rk)
        want = radial + np.cross([0.0, 1.0, 0.0], v) * omega
        assert np.linalg.norm(got - want) <= 1e-4 * np.linalg.norm(want), (a, p, v, got, want)
        assert abs(om - omega) <= 1e-5 * abs(omega) + 1e-12


def _critical_impact_parameter(newton_term, capture_radius=2.3, M=1.0):
    """Critical impact parameter of a unit-speed particle under a = -r_hat (newton_term M / r^2 +
    3 M L^2 / r^4), |v| held at 1 (the continuous limit of the march's per-step normalize(v)):
    scipy DOP853, rtol 1e-10, bisection.  newton_term = 0 is the exact Schwarzschild photon
    equation u'' + u = 3 M u^2 (Binet), whose answer is 3 sqrt(3) M."""
    from scipy.integrate import solve_ivp

    def rhs(t, y):
        p, v = y[:3], y[3:]
        r = np.linalg.norm(p)
        lv = np.cross(p, v)
        a = -(p / r) * (newton_term * M / r ** 2 + 3.0 * M * lv.dot(lv) / r ** 4)
        a = a - a.dot(v) * v
        return np.concatenate([v, a])

    def hit(t, y):
        return np.linalg.norm(y[:3]) - capture_radius
    hit.terminal = True

    def far(t, y):
        return np.linalg.norm(y[:3]) - 200.0
    far.terminal = True
    lo, hi = 3.0, 12.0
    for _ in range(28):
        mid = 0.5 * (lo + hi)
        sol = solve_ivp(rhs, [0.0, 2000.0], np.array([mid, 0.0, -60.0, 0.0, 0.0, 1.0]), method="DOP853",
                        rtol=1e-10, atol=1e-12, events=[hit, far])
        if len(sol.t_events[0]):
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


def test_glsl_march_shadow_edge_is_the_critical_curve_of_its_own_equation_of_motion(oracle):
    """Oracle-independent pin of the whole a16 march (a = 0, far camera, redshift overlay: captured
    rays print black).  The shader's force is -r_hat (M/r^2 + 3 M L^2/r^4) (chunks/metric.ts:
    124-131) with v renormalised every step: the 3 M L^2/r^4 term alone is the exact photon
    equation (critical b = 3 sqrt(3) M = 5.196 M, reproduced below by the same integrator), the
    extra Newtonian M/r^2 term moves the critical curve of the SHADER'S equation to 7.696 M -- the
    reference's GLSL shadow is 48 % wider than Schwarzschild's, a property of the shader, kept.
    The marched edge must sit on that 7.696 M curve within +-2 % (step-size error of the Verlet
    march; measured: captured up to 7.65, escaping from 7.67).  Judged on pixel columns near the
    image centre: rays that stay within |y| < 0.2 of the equatorial plane take 0.3x steps
    (fragment.glsl.ts:160-162) and exhaust the shader's 500-step cap before reaching the horizon
    from b = 6.9 on -- a budget artefact of the shader that the oracle reproduces and this test
    checks as well."""
    assert abs(_critical_impact_parameter(0.0) - 3.0 * np.sqrt(3.0)) < 2e-3   # the integrator itself
    bc = _critical_impact_parameter(1.0)
    assert abs(bc - 7.696) < 5e-3
    W = H = 1024
    SX = 128                                      # every 128th column is enough
    gp = oracle.GlslParams()
    gp.width, gp.height = W, H
    gp.mass, gp.spin, gp.zoom = 1.0, 0.0, 60.0
    gp.mouse[0], gp.mouse[1] = 0.5, 0.5          # no camera rotation: ro = (0, 0, -zoom)
    gp.disk_size, gp.disk_scale_height, gp.disk_density, gp.disk_temp = 50.0, 0.2, 4.0, 9500.0
    gp.lensing_strength, gp.turbulence = 1.0, 0.5
    gp.max_ray_steps, gp.tone_map, gp.quality = 500, 0, 1
    gp.features = 1 | 64                          # LENSING | REDSHIFT
    gp.show_redshift = 1.0
    gp.cam_quat[3] = 1.0
    planes = (np.zeros(65536, np.uint8), np.zeros(65536, np.uint8))
    gp.noise_r, gp.blue_r = planes[0].ctypes.data, planes[1].ctypes.data
    rgba, steps = oracle.glsl_frame(gp, stride=(SX, 1), nthreads=4)
    minres = float(min(W, H))
    xs = (np.arange(0, W, SX) + 0.5 - 0.5 * W) / minres
    ys = ((H - 1 - np.arange(H)) + 0.5 - 0.5 * H) / minres
    cols = np.abs(xs) < 0.13                      # the three columns through the shadow
    captured = (rgba[..., :3] == 0.0).all(axis=-1)[:, cols]
    uv = np.sqrt(xs[None, cols] ** 2 + ys[:, None] ** 2)
    b = 60.0 * np.sin(np.arctan(uv / 1.5))        # |ro x rd| of a pixel
    assert cols.sum() == 3 and captured.any(axis=0).all()
    assert 0.98 * bc <= b[captured].max() <= 1.02 * bc, (b[captured].max(), bc)
    assert 0.98 * bc <= b[~captured].min() <= 1.02 * bc, (b[~captured].min(), bc)
    print("glsl march: captured up to b = %.4f, escaping from b = %.4f (critical curve of the "
          "shader's equation %.4f; Schwarzschild %.4f)"
          % (b[captured].max(), b[~captured].min(), bc, 3.0 * np.sqrt(3.0)))
    # the in-plane row: the 500-step cap ends the march before the horizon from b ~ 6.9 on
    rgba, steps = oracle.glsl_frame(gp, stride=(1, 128), nthreads=4)
    row = 4                                       # iy = 512: uv_y = -0.0005
    xr = (np.arange(W) + 0.5 - 0.5 * W) / minres
    br = 60.0 * np.sin(np.arctan(np.abs(xr) / 1.5))
    cap_row = (rgba[row, :, :3] == 0.0).all(axis=-1)
    assert 6.7 < br[cap_row].max() < 7.2, br[cap_row].max()
    assert (steps[row][cap_row] < 500).all() and (steps[row][~cap_row] == 500).all()


def test_glsl_disk_doppler_factor_equals_the_pinned_kerr_g_factor(L):
    """chunks/disk.ts:78-93 computes delta = 1 / (u^t (1 - Omega L)) from the equatorial metric --
    physics/redshift.rs:65-95 (pinned by redshift.rs:138-171) in f32.  3000 random (r, L, a >= 0):
    5e-5 relative where neither side sits on a clamp."""
    rng = np.random.default_rng(5)
    n = 0
    for _ in range(3000):
        M = 1.0
        spin = float(rng.uniform(0.0, 0.998))
        r = float(np.float32(rng.uniform(6.5, 40.0)))
        lam = float(np.float32(rng.uniform(-4.0, 4.0)))
        ref = L.orc_kerr_g_factor(r, M, spin, lam)
        got = L.orc_hook_glsl_disk_delta(M, spin * M, spin, r, lam)
        if 0.02 < ref < 50.0:
            n += 1
            assert abs(got - ref) <= 5e-5 * ref, (r, lam, spin, got, ref)
    assert n > 2500
    # redshift.rs:138-171 on the shader form
    assert abs(L.orc_hook_glsl_disk_delta(1.0, 0.0, 0.0, 1000.0, 0.0) - 1.0) < 0.01
    assert 0.0 < L.orc_hook_glsl_disk_delta(1.0, 0.0, 0.0, 6.5, 0.0) < 1.0
    assert L.orc_hook_glsl_disk_delta(1.0, 0.0, 0.0, 10.0, 5.0) > L.orc_hook_glsl_disk_delta(1.0, 0.0, 0.0, 10.0, -5.0)


def test_reference_tests_of_the_redshift_potential(L):   # advanced-physics.test.ts:35-75
    M = 1.0
    rs = 2.0 * M
    g = L.orc_hook_glsl_redshift_potential
    assert abs(g(rs, rs)) < 1e-7
    assert abs(g(rs, 1e10) - 1.0) < 1e-5
    assert abs(g(rs, 3 * rs) - np.sqrt(2.0 / 3.0)) < 1e-7
    assert g(rs, rs * 0.5) == 0.0
    vals = [g(rs, r) for r in (rs * 1.01, rs * 2, rs * 5, rs * 10, rs * 100)]
    assert all(b > a for a, b in zip(vals, vals[1:]))
    t_obs = 10000.0 * g(rs, 3 * rs)
    assert 0.0 < t_obs < 10000.0 and abs(t_obs - 10000.0 * np.sqrt(2.0 / 3.0)) < 0.05


def test_reference_tests_of_the_ergosphere_boundary(L):  # advanced-physics.test.ts:77-122
    M, a = 1.0, 0.9
    e = L.orc_hook_glsl_ergosphere_radius
    assert abs(e(M, a, 0.0) - 2.0 * M) < 1e-6
    r_plus = M + np.sqrt(M * M - a * a)
    assert abs(e(M, a, 1.0) - r_plus) < 1e-6 and abs(e(M, a, -1.0) - r_plus) < 1e-6
    assert e(M, a, 0.0) > e(M, a, 1.0)
    for c in (0, 0.3, 0.5, 0.7, 1.0):
        assert abs(e(M, 0.0, c) - 2.0 * M) < 1e-6
    for spin in (0, 0.3, 0.5, 0.7, 0.9, 0.99):
        rh = M + np.sqrt(max(0.0, M * M - spin * spin))
        for c in (0, 0.3, 0.5, 0.7, 1.0):
            assert e(M, spin, c) >= rh - 1e-6
    assert abs(e(M, a, 0.5) - e(M, a, -0.5)) < 1e-7


def test_reference_tests_of_the_isco_polynomial_mirror(L):  # advanced-physics.test.ts:124-167
    """The polynomial of the test file mirrors an older shader; today's metric.ts:19-31 evaluates
    the Bardeen closed form.  Held here: every property the test file asserts that is a property
    of the ISCO itself, plus agreement with the polynomial inside its own fit error (5 %, |a*| <= 0.6)."""
    isco = L.orc_hook_glsl_isco
    M = 1.0
    assert abs(isco(M, 0.0) - 6.0 * M) < 5e-3
    vals = [isco(M, a) for a in (0, 0.2, 0.4, 0.6, 0.8)]
    assert all(b < a for a, b in zip(vals, vals[1:]))
    assert abs(isco(2.0, 1.0) / isco(1.0, 0.5) - 2.0) < 1e-5         # linear in M at fixed a* = 0.5
    for a in np.arange(-1.0, 1.01, 0.1):
        assert isco(M, float(a)) > 0.0
    for spin in (-0.9, -0.5, 0.0, 0.5, 0.9):
        assert isco(M, spin) > M + np.sqrt(max(0.0, M * M - spin * spin))
    for a in (0.0, 0.2, 0.4, 0.6):
        poly = M * (6.0 - 4.627 * a + 2.399 * a * a - 0.772 * a ** 3)
        assert abs(isco(M, a) - poly) < 0.05 * poly, (a, isco(M, a), poly)   # the fit is good to ~4 %


def test_reference_tests_of_doppler_beaming(L):          # advanced-physics.test.ts:169-195
    """Exponent: the shader's disk block uses delta^3.5 (chunks/disk.ts:95; the test file argues for
    3 + alpha): the hook must be that power, floored at 0.01, and keep the test file's inequalities."""
    bm = L.orc_hook_glsl_beaming
    assert bm(1.5) > 1.0 and bm(0.8) < 1.0 and bm(1.0) == 1.0
    for d in (0.3, 0.8, 1.0, 1.5, 2.7):
        assert abs(bm(d) - d ** 3.5) <= 2e-6 * d ** 3.5
    assert bm(0.1) == np.float32(0.01)


def test_reference_test_of_the_verlet_update(L):         # advanced-physics.test.ts:196-240
    """The march's two update lines (fragment.glsl.ts:176-191) on x'' = -x, dt 0.01, 10^4 steps:
    |E_final - E_0| < 1e-4, as the reference asserts of the same two lines."""
    x, v = C.c_float(1.0), C.c_float(0.0)
    L.orc_hook_glsl_verlet_oscillator(C.byref(x), C.byref(v), 1.0, 0.01, 10000)
    e = 0.5 * v.value ** 2 + 0.5 * x.value ** 2
    assert abs(e - 0.5) < 1e-4, e
    # and the phase: 100 time units of cos(t), to the integrator's O(dt^2) phase error
    assert abs(x.value - np.cos(100.0)) < 2e-3 and abs(v.value + np.sin(100.0)) < 2e-3


def test_glsl_blackbody_fit_against_its_literal_anchor_values(L):
    """chunks/blackbody.ts:9-34 (the Tanner Helland fit): piecewise anchors computed by hand from
    the literal constants."""
    def bb(t):
        o = np.zeros(3, np.float32)
        L.orc_hook_glsl_blackbody(t, P(o))
        return o.astype(np.float64)
    # 6600 K: t = 66 -> r = 255, g = 99.4708025861 ln 66 - 161.1195681661, b = 138.5177312231 ln 56 - 305.0447927307
    g = 99.4708025861 * np.log(66.0) - 161.1195681661
    b = 138.5177312231 * np.log(56.0) - 305.0447927307
    want = np.array([1.0, (g / 255.0) ** 2.2, (b / 255.0) ** 2.2])
    assert np.allclose(bb(6600.0), want, rtol=2e-5)
    # 1500 K: t = 15 <= 19 -> b = 0
    g = 99.4708025861 * np.log(15.0) - 161.1195681661
    assert np.allclose(bb(1500.0), [1.0, (g / 255.0) ** 2.2, 0.0], rtol=2e-5, atol=1e-9)
    # 10000 K: t = 100 -> r, g power laws of (t - 60), b = 255
    r = 329.698727446 * 40.0 ** -0.1332047592
    g = 288.1221695283 * 40.0 ** -0.0755148492
    assert np.allclose(bb(10000.0), [(r / 255.0) ** 2.2, (g / 255.0) ** 2.2, 1.0], rtol=2e-5)


# --------------------------------------------------------------------------------------------
# 8f-4: post chain against the shaders' literal constants on hand-computed images
# --------------------------------------------------------------------------------------------
YCOCG = np.array([[0.25, 0.5, 0.25], [0.5, 0.0, -0.5], [-0.25, 0.5, -0.25]])


def _to_rgb(y):
    return np.array([y[0] + y[1] - y[2], y[0] + y[2], y[0] - y[1] - y[2]])


def _neigh(img, px, py):
    h, w, _ = img.shape
    return [img[min(max(py + dy, 0), h - 1), min(max(px + dx, 0), w - 1), :3]
            for dy in (-1, 0, 1) for dx in (-1, 0, 1)]


def _hand_resolve(cur, hist_rgb, px, py, k_sigma, alpha_of_sd):
    s = np.array([YCOCG @ t for t in _neigh(cur, px, py)])
    mean = s.sum(0) / 9.0
    sd = np.sqrt(np.maximum((s * s).sum(0) / 9.0 - mean * mean, 0.0))
    hy = np.clip(YCOCG @ hist_rgb, mean - k_sigma * sd, mean + k_sigma * sd)
    cy = YCOCG @ cur[py, px, :3]
    a = alpha_of_sd(sd)
    return _to_rgb(cy * (1.0 - a) + hy * a), sd


def test_taa_variance_clip_on_hand_computed_3x3(oracle):
    """reprojection.glsl.ts:76-113: box = mean +- 1.5 sigma over the 3x3 YCoCg neighbourhood,
    weight = 1 - clamp(4 sigma_Y, 0, 0.55), alpha = moving ? 0 : blend * weight."""
    rng = np.random.default_rng(11)
    cur = np.zeros((3, 3, 4), np.float32)
    cur[..., :3] = rng.uniform(0.0, 1.0, (3, 3, 3)).astype(np.float32)
    cur[..., 3] = 1.0
    for hist_val in ((0.2, 0.4, 0.1), (5.0, 5.0, 5.0), (0.0, 0.0, 0.0)):   # inside / above / below the box
        hist = np.zeros((3, 3, 4), np.float32)
        hist[..., :3] = hist_val
        out = oracle.taa_resolve(cur, hist, blend_factor=0.75, camera_moving=False, half_storage=False)
        for (px, py) in ((1, 1), (0, 0), (2, 1)):     # centre, corner, edge (CLAMP_TO_EDGE taps)
            want, sd = _hand_resolve(cur.astype(np.float64), np.array(hist_val), px, py, 1.5,
                                     lambda sd: 0.75 * (1.0 - min(max(sd[0] * 4.0, 0.0), 0.55)))
            assert np.allclose(out[py, px, :3], want, atol=2e-6), (hist_val, px, py, out[py, px], want)
    # flat neighbourhood: sigma = 0, the box collapses onto the mean, any history is clipped to it
    flat = np.zeros((3, 3, 4), np.float32)
    flat[..., :3] = (0.3, 0.6, 0.2)
    wild = np.zeros((3, 3, 4), np.float32)
    wild[..., :3] = (9.0, -3.0, 4.0)
    out = oracle.taa_resolve(flat, wild, half_storage=False)
    assert np.allclose(out[..., :3], flat[..., :3], atol=1e-6)
    # camera moving: alpha = 0, the current frame passes through
    out = oracle.taa_resolve(cur, wild, camera_moving=True, half_storage=False)
    assert np.allclose(out[..., :3], cur[..., :3], atol=1e-6)
    # RGBA16F target: every stored channel is a binary16 value
    out = oracle.taa_resolve(cur, hist, half_storage=True)
    assert np.array_equal(out[..., :3], out[..., :3].astype(np.float16).astype(np.float32))


def test_ataa_variance_clip_on_hand_computed_3x3(oracle):
    """ataa.wgsl.ts:29-86: texel loads clamped to the image, box = mean +- 2 sigma, feedback 0.92.
    A constant history makes the reprojected tap independent of the camera."""
    rng = np.random.default_rng(12)
    cur = np.zeros((3, 3, 4), np.float32)
    cur[..., :3] = rng.uniform(0.0, 1.0, (3, 3, 3)).astype(np.float32)
    cur[..., 3] = 1.0
    cam = oracle.AtaaCamera()
    for k in (0, 5, 10, 15):
        cam.inv_view[k] = cam.inv_proj[k] = cam.prev_view_proj[k] = 1.0
    cam.position[2] = 5.0
    for hist_val in ((0.3, 0.3, 0.3), (4.0, 0.0, 0.0)):
        hist = np.zeros((3, 3, 4), np.float32)
        hist[..., :3] = hist_val
        out = oracle.ataa_resolve(cam, cur, hist, half_storage=False)
        for (px, py) in ((1, 1), (0, 2), (1, 0)):
            want, _ = _hand_resolve(cur.astype(np.float64), np.array(hist_val), px, py, 2.0, lambda sd: 0.92)
            assert np.allclose(out[py, px, :3], want, atol=2e-6), (hist_val, px, py, out[py, px], want)


def test_bloom_constants_on_hand_computed_images(oracle):
    """bloom.glsl.ts:35-127 + bloom.ts:23-39 (threshold 0.8, intensity 0.5, 2 passes):
    luminance (0.299, 0.587, 0.114), 9-tap weights (0.227027, 0.1945946, 0.1216216, 0.054054,
    0.016216), ACES (2.51, 0.03, 2.43, 0.59, 0.14), gamma 0.4545 -- on constant images, where
    every bilinear tap returns the constant and the blur multiplies by the weight sum."""
    wsum = 0.227027 + 2.0 * (0.1945946 + 0.1216216 + 0.054054 + 0.016216)

    def aces(c):
        return np.clip((c * (2.51 * c + 0.03)) / (c * (2.43 * c + 0.59) + 0.14), 0.0, 1.0)

    def run(rgb, **kw):
        img = np.zeros((16, 16, 4), np.float32)
        img[..., :3] = rgb
        img[..., 3] = 1.0
        return oracle.bloom(img, half_storage=False, **kw)

    for rgb in ((2.0, 2.0, 2.0), (0.9, 0.9, 0.7), (3.0, 0.0, 0.0)):       # luminance > 0.8: blooms
        c = np.array(rgb)
        assert c @ [0.299, 0.587, 0.114] > 0.8
        want = aces(c + c * wsum ** 4 * 0.5) ** 0.4545                      # 2 passes x (H, V)
        out = run(rgb)
        assert np.allclose(out[..., :3], want, atol=3e-6), (rgb, out[8, 8], want)
    for rgb in ((0.5, 0.5, 0.5), (1.0, 0.0, 0.0), (0.0, 0.0, 7.0)):       # luminance <= 0.8: no bloom
        c = np.array(rgb)
        assert c @ [0.299, 0.587, 0.114] <= 0.8
        out = run(rgb)
        assert np.allclose(out[..., :3], aces(c) ** 0.4545, atol=3e-6), (rgb, out[8, 8])
    # the BloomConfig knobs
    c = np.array((2.0, 1.0, 0.5))
    out = run(tuple(c), threshold=0.5, intensity=1.0, blur_passes=1)
    assert np.allclose(out[..., :3], aces(c + c * wsum ** 2 * 1.0) ** 0.4545, atol=3e-6)
    out = run(tuple(c), blur_passes=0)                                      # no blur: bloom = bright pass
    assert np.allclose(out[..., :3], aces(c + c * 0.5) ** 0.4545, atol=3e-6)


def test_bloom_blur_kernel_on_an_impulse(oracle):
    """One lit quarter-resolution texel.  Scene 64x64, a 4x4 block (= one texel of the 1/4-size blur
    targets = 2x2 texels of the 1/2-size bright target) far above the threshold.  After one (H, V)
    pass the blur target is 100 w[k] w[m] around it (every tap of bloom.glsl.ts:79-85 lands on the
    middle of a 2x2 group of bright texels), so its row sums are B[m] = 100 w[m] sum(w).  The combine
    pass reads the blur target bilinearly at full-resolution pixel centres, i.e. at quarter-texel
    offsets -0.375, -0.125, +0.125, +0.375: four full rows of quarter row j pick up
    3 B[j] + 0.5 B[j-1] + 0.5 B[j+1], and the sum along x is preserved (x 4).  Hand numbers from the
    literal weights, compared after undoing ACES and gamma, away from the lit block."""
    w = np.array([0.016216, 0.054054, 0.1216216, 0.1945946, 0.227027, 0.1945946, 0.1216216, 0.054054, 0.016216])
    S = np.zeros((64, 64, 4), np.float32)
    S[..., 3] = 1.0
    S[32:36, 32:36, :3] = 100.0
    out = oracle.bloom(S, threshold=0.8, intensity=1.0, blur_passes=1, half_storage=False)

    def inv(o):  # undo gamma, then ACES (monotone below saturation): c = x(2.51x+.03)/(x(2.43x+.59)+.14)
        c = o.astype(np.float64) ** (1.0 / 0.4545)
        a_, b_, c_ = 2.51 - 2.43 * c, 0.03 - 0.59 * c, -0.14 * c
        return (-b_ + np.sqrt(b_ * b_ - 4 * a_ * c_)) / (2 * a_)
    row_q = np.array([inv(out[4 * j:4 * j + 4, :, 0]).sum() / 16.0 for j in range(16)])
    B = np.zeros(18)                                  # B[j + 1] = row sum of quarter row j
    for j in range(16):
        k = j - 8 + 4
        if 0 <= k < 9:
            B[j + 1] = 100.0 * w[k] * w.sum()
    want = np.array([(3.0 * B[j + 1] + 0.5 * (B[j] + B[j + 2])) / 4.0 for j in range(16)])
    mask = np.ones(16, bool)
    mask[7:10] = False            # the lit block and its bilinear neighbours: scene + bloom saturates ACES
    assert np.allclose(row_q[mask], want[mask], rtol=5e-4, atol=1e-4), (row_q, want)
    assert row_q[:3].max() == 0.0 and row_q[14:].max() == 0.0      # 9 taps: nothing beyond +-4 texels (+1 bilinear)


# --------------------------------------------------------------------------------------------
# the f64 twin of the compute march (oracle/wgsl_f64_twin.c), the instrument behind
# tests/measure_c4_budget_rays.py: it must be the SAME discrete march as the f32 oracle
# --------------------------------------------------------------------------------------------
def test_f64_twin_of_the_compute_march_is_the_same_algorithm(oracle):
    """Every 24th pixel of the 1080p / 512-step frame at a = 0.999 (stars off): the double-precision
    twin takes the same number of steps as the f32 shader-order oracle on > 99.9 % of the pixels
    (the rest: rays on the unstable photon orbit, where f32 rounding decides the step count -- the
    question the twin exists to answer), never more than the budget, and its colours sit within
    2e-3 of peak of the f32 ones at the 99.9th percentile."""
    W, H = 1920, 1080
    th = np.deg2rad(97.0)
    cam = oracle.camera_look_at((60.0 * np.sin(th), 60.0 * np.cos(th), 0.0), aspect=W / H)
    op = oracle.WgslParams()
    for k in range(16):
        op.inv_view[k], op.inv_proj[k] = cam.inv_view[k], cam.inv_proj[k]
    for k in range(3):
        op.position[k] = cam.position[k]
    op.mass, op.spin, op.width, op.height, op.max_steps, op.stars = 1.0, 0.999, W, H, 512, 0
    rgba, steps = oracle.wgsl_frame(op, stride=(24, 24), nthreads=8)
    ys, xs = np.mgrid[0:H:24, 0:W:24]
    d = oracle.wgsl_pixels_f64(op, np.stack([xs.ravel(), ys.ravel()], 1), nthreads=8)
    ds = np.abs(d["steps"].astype(np.int64) - steps.ravel().astype(np.int64))
    assert (ds == 0).mean() > 0.999 and d["steps"].max() <= 512
    assert set(np.unique(d["cls"])) <= {0, 1, 2, 3} and (d["cls"] == 1).mean() > 0.5
    peak = rgba[..., :3].max()
    dc = np.abs(d["rgb"] - rgba.reshape(-1, 4)[:, :3]).max(1) / peak
    assert np.percentile(dc, 99.9) <= 2e-3
    # exit classes are consistent with the step counts
    assert ((d["cls"] == 2) == (d["steps"] == 512)).all()


def test_lattice_form_of_the_noise_hash_is_the_shaders_arithmetic():
    """csrc/glsl_fragment.hpp glsl_noise_lattice (FAST) indexes the noise texel of a cell corner by
    integer arithmetic: ((x + 37 z) mod 256, (y + 37 z) mod 256).  The shader's hash() (noise.ts:3-9)
    reaches its texel through f32: uv = p.xy + p.z * 37, texture coordinate (uv + 0.5) / 256, LINEAR
    filtering at s * 256 - 0.5.  For integer corners with |coordinates| < 2^17 every one of those f32
    operations is exact: the filter position is the integer uv itself (fractional weights exactly 0 --
    one texel) and its REPEAT-wrapped index is the integer formula.  Checked here in numpy f32, with the
    products formed both ways a compiler may form them (separate multiply and add, or one fma)."""
    rng = np.random.default_rng(4)
    n = 200000
    lim = 2 ** 17 - 1
    pts = rng.integers(-lim, lim + 1, size=(n, 3))
    pts[:64] = rng.choice([-lim, lim, 0, -1, 1, 255, 256, -256, -255], size=(64, 3))
    x, y, z = (pts[:, k].astype(np.float32) for k in range(3))
    for fused in (False, True):
        if fused:   # fma(z, 37, x): one rounding of the exact value -- exact integers below 2^24 either way
            uvx = (pts[:, 0].astype(np.float64) + pts[:, 2].astype(np.float64) * 37.0).astype(np.float32)
            uvy = (pts[:, 1].astype(np.float64) + pts[:, 2].astype(np.float64) * 37.0).astype(np.float32)
        else:
            uvx = x + z * np.float32(37.0)
            uvy = y + z * np.float32(37.0)
        s = (uvx + np.float32(0.5)) / np.float32(256.0)
        t = (uvy + np.float32(0.5)) / np.float32(256.0)
        u = s * np.float32(256.0) - np.float32(0.5)
        v = t * np.float32(256.0) - np.float32(0.5)
        fu, fv = np.floor(u), np.floor(v)
        assert np.all(u - fu == 0) and np.all(v - fv == 0)            # a texel centre: weights (1, 0, 0, 0)
        ix = fu.astype(np.int64) & 255
        iy = fv.astype(np.int64) & 255
        assert np.array_equal(ix, (pts[:, 0] + 37 * pts[:, 2]) & 255)
        assert np.array_equal(iy, (pts[:, 1] + 37 * pts[:, 2]) & 255)
    # the kernel reduces each coordinate mod 256 first (i - 256 floor(i / 256), exact in f32 for any float):
    # the index is unchanged
    r = (x - np.float32(256.0) * np.floor(x * np.float32(0.00390625))).astype(np.int64)
    rz = (z - np.float32(256.0) * np.floor(z * np.float32(0.00390625))).astype(np.int64)
    assert np.all((r >= 0) & (r < 256))
    assert np.array_equal((r + 37 * rz) & 255, (pts[:, 0] + 37 * pts[:, 2]) & 255)
