"""Oracle checks that do not depend on the restatement itself:
exact conservation of E and L_z, Carter-Q / H drift, Bardeen critical impact
parameters, and a scipy DOP853 integration of independently written Hamilton
equations.  The reference pins none of these endpoints ("parity unpinned")."""
import math

import numpy as np
import pytest

PI_2 = math.pi / 2


def test_energy_and_lz_exactly_conserved(oracle):
    m = oracle.metric(oracle.KERR_KS, 1.0, 0.9)
    v = [0, 30.0, 1.2, 0.3, -1.0, -0.9, 1.5, 2.5]
    t, path = oracle.integrate_path(v, m, oracle.options(max_steps=3000))
    assert len(path) == t.steps_taken + 1
    assert np.all(path[:, 4] == -1.0) and np.all(path[:, 7] == 2.5)  # geodesic/hamiltonian.rs:33


@pytest.mark.parametrize("kind", [0, 1])
def test_carter_constant_and_hamiltonian_drift(oracle, kind):
    import ctypes as C
    m = oracle.metric(kind, 1.0, 0.9)
    v = [0, 30.0, 1.2, 0.3, -1.0, -0.9, 1.5, 2.5]
    t, path = oracle.integrate_path(v, m, oracle.options(max_steps=3000))
    L = oracle.lib()
    q = []
    for row in path[1:]:
        s = oracle.make_state(row)
        q.append(L.orc_carter_constant(C.byref(s), C.byref(m)))
    q = np.array(q)
    assert np.max(np.abs(q - q[0])) / abs(q[0]) < 1e-5
    assert t.max_hamiltonian_drift < 1e-5


def _classify(oracle, kind, spin, lz, r0=50.0):
    m = oracle.metric(kind, 1.0, spin)
    t = oracle.integrate([0, r0, PI_2, 0, -1.0, -1.0, 0.0, lz], m, oracle.options(max_steps=20000))
    return t.termination


def test_schwarzschild_critical_impact_parameter(oracle):
    # b_crit = 3 sqrt(3) M = 5.196 (physics/shadow.rs:39-59 uses the same curve)
    for kind in (oracle.KERR_KS, oracle.KERR_BL):
        assert _classify(oracle, kind, 0.0, 5.10) == oracle.TERM_HORIZON
        assert _classify(oracle, kind, 0.0, 5.30) == oracle.TERM_ESCAPE


def test_schwarzschild_struct_is_bug_compatible(oracle):
    """metric/schwarzschild.rs:87 has d(g^tt)/dr = -2M/(r^2 f^2); the derivative of
    g^tt = -1/f is +2M/(r^2 f^2).  The oracle restates the reference as written, so
    the `Schwarzschild` metric does NOT conserve H (the a = 0 Kerr metrics do).
    The a = 0 plumbing config therefore uses Kerr-Schild with spin 0."""
    v = [0, 50.0, PI_2, 0, -1.0, -1.0, 0.0, 5.3]
    bad = oracle.integrate(v, oracle.metric(oracle.SCHWARZSCHILD, 1.0, 0.0), oracle.options())
    good = oracle.integrate(v, oracle.metric(oracle.KERR_BL, 1.0, 0.0), oracle.options())
    assert good.max_hamiltonian_drift < 1e-6
    assert bad.max_hamiltonian_drift > 1e-3
    dr, _ = oracle.hamiltonian_derivatives(oracle.metric(oracle.SCHWARZSCHILD, 1.0, 0.0), 10.0,
                                           PI_2, [-1.0, 0.0, 0.0, 0.0])
    assert dr == 0.5 * (-2.0 / (100.0 * 0.8 * 0.8))  # sign as in the reference


def test_kerr_equatorial_critical_impact_parameters(oracle):
    a = 0.9  # Bardeen 1973: b_c(+) = 6 cos(acos(-a)/3) - a ; b_c(-) = -(6 cos(acos(a)/3) + a)
    b_pro = 6.0 * math.cos(math.acos(-a) / 3.0) - a
    b_ret = -(6.0 * math.cos(math.acos(a) / 3.0) + a)
    assert abs(b_pro - 2.8449) < 1e-3 and abs(b_ret + 6.8323) < 1e-3
    for kind in (oracle.KERR_KS, oracle.KERR_BL):
        assert _classify(oracle, kind, a, b_pro - 0.08) == oracle.TERM_HORIZON
        assert _classify(oracle, kind, a, b_pro + 0.08) == oracle.TERM_ESCAPE
        assert _classify(oracle, kind, a, b_ret + 0.08) == oracle.TERM_HORIZON
        assert _classify(oracle, kind, a, b_ret - 0.08) == oracle.TERM_ESCAPE


def _ks_hamiltonian(r, th, p, M, a):
    """H = 1/2 g^{mu nu} p_mu p_nu, ingoing Kerr-Schild, written independently (works on
    complex r / th for complex-step differentiation)."""
    pt, pr, pth, pph = p
    s2 = np.sin(th) ** 2
    sig = r * r + a * a * (1.0 - s2)
    dl = r * r - 2.0 * M * r + a * a
    gtt = -(1.0 + 2.0 * M * r / sig)
    gtr = 2.0 * M * r / sig
    return 0.5 * (gtt * pt * pt + 2 * gtr * pt * pr + dl / sig * pr * pr + pth * pth / sig +
                  pph * pph / (sig * s2) + 2 * (a / sig) * pr * pph)


def _rhs(lam, y, M, a, pt, pph):
    t, r, th, ph, pr, pth = y
    p = (pt, pr, pth, pph)
    h = 1e-30
    dHdr = np.imag(_ks_hamiltonian(r + 1j * h, th, p, M, a)) / h
    dHdth = np.imag(_ks_hamiltonian(r, th + 1j * h, p, M, a)) / h
    dHdpr = np.imag(_ks_hamiltonian(r, th, (pt, pr + 1j * h, pth, pph), M, a)) / h
    dHdpth = np.imag(_ks_hamiltonian(r, th, (pt, pr, pth + 1j * h, pph), M, a)) / h
    dHdpt = np.imag(_ks_hamiltonian(r, th, (pt + 1j * h, pr, pth, pph), M, a)) / h
    dHdpph = np.imag(_ks_hamiltonian(r, th, (pt, pr, pth, pph + 1j * h), M, a)) / h
    return [dHdpt, dHdpr, dHdpth, dHdpph, -dHdr, -dHdth]


@pytest.mark.parametrize("v", [
    [0, 20.0, PI_2, 0, -1.0, -1.0, 0.0, 3.5],
    [0, 40.0, 1.1, 0.5, -1.0, -1.0, 2.0, -9.0],
    [0, 25.0, 2.0, 1.0, -1.0, -0.8, -1.0, 4.2],
])
def test_endpoint_against_independent_dop853(oracle, v):
    from scipy.integrate import solve_ivp
    import ctypes as C
    M, a = 1.0, 0.9
    m = oracle.metric(oracle.KERR_KS, M, a)
    tr = oracle.integrate(v, m, oracle.options(tolerance=1e-10, max_steps=20000))
    assert tr.termination == oracle.TERM_ESCAPE
    fs = oracle.state_to_list(tr.final_state)
    s0 = oracle.make_state(v)
    oracle.lib().orc_renormalize_null(C.byref(s0), C.byref(m))
    y0 = [s0.x[0], s0.x[1], s0.x[2], s0.x[3], s0.p[1], s0.p[2]]

    def hit(lam, y, *args):
        return y[1] - fs[1]
    hit.terminal, hit.direction = True, 1.0
    sol = solve_ivp(_rhs, [0.0, 1e5], y0, method="DOP853", rtol=1e-12, atol=1e-12,
                    events=hit, args=(M, a, v[4], v[7]))
    assert sol.status == 1
    ye = sol.y_events[0][0]
    # same geodesic, compared at the same radius
    assert abs(ye[0] - fs[0]) < 1e-5 * max(1.0, abs(fs[0]))  # t
    assert abs(ye[2] - fs[2]) < 1e-6                          # theta
    assert abs(ye[3] - fs[3]) < 1e-6                          # phi
    assert abs(ye[4] - fs[5]) < 1e-6                          # p_r
    assert abs(ye[5] - fs[6]) < 1e-5                          # p_theta


def test_symplectic_and_rk4_agree_with_rkf45(oracle):
    m = oracle.metric(oracle.KERR_KS, 1.0, 0.5)
    v = [0, 12.0, 1.3, 0.0, -1.0, -1.0, 0.5, 3.9]
    n = 400
    a = oracle.integrate(v, m, oracle.options(method=oracle.METHOD_RK4, step_size=0.01, max_steps=n,
                                              renormalize_interval=0))
    b = oracle.integrate(v, m, oracle.options(method=oracle.METHOD_SYMPLECTIC, step_size=0.01,
                                              max_steps=n, renormalize_interval=0))
    xa, xb = np.array(oracle.state_to_list(a.final_state)), np.array(oracle.state_to_list(b.final_state))
    assert a.steps_taken == n and b.steps_taken == n
    assert np.max(np.abs(xa - xb)) < 1e-4


def test_fehlberg_tableau_orders(oracle):
    """Oracle-independent pin of the stepper's tableau (integrator.rs:113-190): against an
    independently integrated solution (scipy DOP853 on the complex-step Hamilton equations above)
    the 5th-order update's local error falls as h^6 and the embedded 4(5) estimate as h^5.
    A mistyped node or weight breaks the orders."""
    from scipy.integrate import solve_ivp
    import ctypes as C
    M, a = 1.0, 0.9
    m = oracle.metric(oracle.KERR_KS, M, a)
    v = [0, 12.0, 1.2, 0.3, -1.0, -0.7, 1.5, 3.0]
    s0 = oracle.make_state(v)
    oracle.lib().orc_renormalize_null(C.byref(s0), C.byref(m))
    y0 = [s0.x[0], s0.x[1], s0.x[2], s0.x[3], s0.p[1], s0.p[2]]
    errs, ests = [], []
    hs = [0.8, 0.4, 0.2]
    for h in hs:
        out = oracle.State()
        est = oracle.lib().orc_rkf45_step(C.byref(s0), C.byref(m), h, C.byref(out))
        sol = solve_ivp(_rhs, [0.0, h], y0, method="DOP853", rtol=1e-13, atol=1e-15, args=(M, a, v[4], v[7]))
        ye = sol.y[:, -1]
        got = [out.x[0], out.x[1], out.x[2], out.x[3], out.p[1], out.p[2]]
        errs.append(max(abs(g - e) for g, e in zip(got, ye)))
        ests.append(est)
    order_sol = [np.log2(errs[i] / errs[i + 1]) for i in range(2)]
    order_est = [np.log2(ests[i] / ests[i + 1]) for i in range(2)]
    assert all(5.3 < o < 6.8 for o in order_sol), (errs, order_sol)     # local error O(h^6)
    assert all(4.4 < o < 5.6 for o in order_est), (ests, order_est)     # estimate O(h^5)
    # the estimate uses the coordinates only and bounds their true 4th-order error from above-ish
    assert ests[-1] > errs[-1]


def test_step_controller_constants(oracle):
    """AdaptiveStepper (integrator.rs:53-107) pinned against its documented constants, with the
    expected next step written here from SURVEY a10: safety 0.9, exponents -1/5 (accept) and -1/4
    (reject), growth cap 5 (and 5 outright below ratio 1e-4), shrink floor 0.1, |h| <= 10,
    forced step at 1e-5."""
    import ctypes as C
    L = oracle.lib()
    m = oracle.metric(oracle.KERR_KS, 1.0, 0.9)
    v = [0, 12.0, 1.2, 0.3, -1.0, -0.7, 1.5, 3.0]

    def fresh():
        s = oracle.make_state(v)
        L.orc_renormalize_null(C.byref(s), C.byref(m))
        return s

    def err_of(h):
        out = oracle.State()
        return L.orc_rkf45_step(C.byref(fresh()), C.byref(m), h, C.byref(out)), out

    tries = C.c_uint64(0)
    # accepted at once: next = h * min(0.9 * ratio^-0.2, 5), clamped to 10
    for h, tol in ((0.2, 1e-8), (0.05, 1e-8), (0.4, 1e-5), (3.0, 1e-1)):
        err, out = err_of(h)
        ratio = err / tol
        assert ratio <= 1.0
        growth = 5.0 if ratio < 1e-4 else 0.9 * ratio ** -0.2
        want = max(-10.0, min(10.0, h * min(growth, 5.0)))
        s = fresh()
        tries.value = 0
        nxt = L.orc_adaptive_step(C.byref(s), C.byref(m), h, tol, C.byref(tries))
        assert tries.value == 1 and abs(nxt - want) <= 1e-15 * abs(want)
        assert oracle.state_to_list(s) == oracle.state_to_list(out)      # the 5th-order candidate was taken
    # h_try beyond the cap is clamped before the first try
    s = fresh()
    L.orc_adaptive_step(C.byref(s), C.byref(m), 50.0, 1e30, C.byref(tries))
    _, out10 = err_of(10.0)
    assert oracle.state_to_list(s) == oracle.state_to_list(out10)
    # rejected once: h <- h * max(0.9 * ratio^-0.25, 0.1), then accepted from there
    h, tol = 0.8, 1e-9
    err, _ = err_of(h)
    assert err / tol > 1.0
    h2 = h * max(0.9 * (err / tol) ** -0.25, 0.1)
    err2, out2 = err_of(h2)
    if err2 / tol <= 1.0:
        s = fresh()
        tries.value = 0
        L.orc_adaptive_step(C.byref(s), C.byref(m), h, tol, C.byref(tries))
        assert tries.value == 2 and oracle.state_to_list(s) == oracle.state_to_list(out2)
    # a tolerance nothing can meet: the step is forced at 1e-5 and handed back as the next h
    s = fresh()
    tries.value = 0
    nxt = L.orc_adaptive_step(C.byref(s), C.byref(m), 0.5, 1e-300, C.byref(tries))
    _, outmin = err_of(1e-5)
    assert nxt == 1e-5 and oracle.state_to_list(s) == oracle.state_to_list(outmin) and tries.value >= 5
