"""Pins the CPU oracle against every known-answer test the reference holds on or
near the geodesic path.  Each test names the reference
test it restates (paths under /root/reference/physics-engine)."""
import math

import numpy as np
import pytest

PI_2 = math.pi / 2


def test_schwarzschild_isco(oracle):  # gravitas-core/src/metric/kerr.rs:507-516
    m = oracle.metric(oracle.KERR_BL, 1.0, 0.0)
    assert abs(oracle.lib().orc_isco(m, 0) - 6.0) < 1e-6


def test_extreme_kerr_isco(oracle):  # kerr.rs:518-528
    m = oracle.metric(oracle.KERR_BL, 1.0, 0.998)
    assert oracle.lib().orc_isco(m, 0) < 1.5


def test_event_horizon(oracle):  # kerr.rs:530-543
    L = oracle.lib()
    assert abs(L.orc_event_horizon(oracle.metric(oracle.KERR_BL, 1.0, 0.0)) - 2.0) < 1e-12
    assert abs(L.orc_event_horizon(oracle.metric(oracle.KERR_BL, 1.0, 1.0)) - 1.0) < 1e-12


def test_photon_sphere(oracle):  # kerr.rs:545-554
    assert abs(oracle.lib().orc_photon_sphere(oracle.metric(oracle.KERR_BL, 1.0, 0.0)) - 3.0) < 1e-6


def test_metric_signature(oracle):  # kerr.rs:556-566
    g = oracle.covariant(oracle.metric(oracle.KERR_BL, 1.0, 0.5), 10.0, PI_2)
    assert g[0] < 0 and g[5] > 0 and g[10] > 0 and g[15] > 0


def test_hamiltonian_consistency_bl_vs_ks(oracle):  # kerr.rs:568-597
    bl = oracle.metric(oracle.KERR_BL, 1.0, 0.5)
    ks = oracle.metric(oracle.KERR_KS, 1.0, 0.5)
    r, a = 5.0, 0.5
    p_bl = np.array([-1.0, 0.0, 0.0, 2.0])
    delta = r * r - 2.0 * r + a * a
    p_r_ks = p_bl[1] + (2.0 * r * 1.0 - a * p_bl[3]) / delta
    p_ks = np.array([p_bl[0], p_r_ks, p_bl[2], p_bl[3]])
    L = oracle.lib()
    h_bl = 0.5 * L.orc_contract(oracle._ptr(oracle.contravariant(bl, r, PI_2)), oracle._ptr(p_bl))
    h_ks = 0.5 * L.orc_contract(oracle._ptr(oracle.contravariant(ks, r, PI_2)), oracle._ptr(p_ks))
    assert abs(h_bl - h_ks) < 1e-8


def test_kerr_doctest(oracle):  # kerr.rs:28-34
    m = oracle.metric(oracle.KERR_BL, 1.0, 0.9)
    assert abs(oracle.lib().orc_event_horizon(m) - 1.4358898944) < 1e-6
    assert abs(oracle.lib().orc_isco(m, 0) - 2.3209) < 0.01


def test_spin_clamped(oracle):  # kerr.rs:48-54
    assert oracle.metric(oracle.KERR_BL, 1.0, 1.7).spin == 1.0
    assert oracle.metric(oracle.KERR_KS, 1.0, -3.0).spin == -1.0


def test_g_factor_at_infinity(oracle):  # gravitas-core/src/physics/redshift.rs:138-147
    assert abs(oracle.lib().orc_kerr_g_factor(1000.0, 1.0, 0.0, 0.0) - 1.0) < 0.01


def test_g_factor_redshift_near_isco(oracle):  # redshift.rs:149-158
    g = oracle.lib().orc_kerr_g_factor(6.5, 1.0, 0.0, 0.0)
    assert 0.0 < g < 1.0


def test_g_factor_blueshift_approaching(oracle):  # redshift.rs:160-171
    L = oracle.lib()
    assert L.orc_kerr_g_factor(10.0, 1.0, 0.0, 5.0) > L.orc_kerr_g_factor(10.0, 1.0, 0.0, -5.0)


def test_metric_tensor_contract_and_layout(oracle):  # tensor/metric_tensor.rs:112-133
    g = np.zeros(16)
    g[0], g[5], g[10], g[15] = -1.0, 1.0, 1.0, 1.0
    p = np.array([1.0, 2.0, 3.0, 4.0])
    assert oracle.lib().orc_contract(oracle._ptr(g), oracle._ptr(p)) == -1 + 4 + 9 + 16


def test_integrate_doctest_runs(oracle):  # geodesic/mod.rs:169-179 (asserts nothing upstream)
    m = oracle.metric(oracle.KERR_BL, 1.0, 0.9)
    t = oracle.integrate([0, 20.0, PI_2, 0, -1.0, -1.0, 0.0, 3.5], m, oracle.options())
    assert t.termination == oracle.TERM_ESCAPE and t.final_state.x[1] > 1000.0
    assert t.max_hamiltonian_drift < 1e-6


def test_legacy_hamiltonian_drift_audit(oracle):  # _legacy_src/integrator.rs:102-150
    m = oracle.metric(oracle.KERR_BL, 1.0, 0.9)
    s = oracle.make_state([0.0, 20.0, 1.57, 0.0, -1.0, -1.0, 0.0, 3.5])
    import ctypes as C
    L = oracle.lib()
    L.orc_renormalize_null(C.byref(s), C.byref(m))
    h, max_drift = 0.05, 0.0
    for _ in range(5000):
        h = L.orc_adaptive_step(C.byref(s), C.byref(m), h, 1e-8, None)
        max_drift = max(max_drift, abs(L.orc_hamiltonian(C.byref(s), C.byref(m))))
        if s.x[1] < 2.1:
            break
    assert max_drift < 1e-4


def test_legacy_horizon_crossing(oracle):  # _legacy_src/integrator.rs:352-386
    import ctypes as C
    m = oracle.metric(oracle.KERR_KS, 1.0, 0.9)
    s = oracle.make_state([0.0, 3.0, 1.57, 0.0, -1.0, -1.0, 0.0, 0.0])
    L = oracle.lib()
    L.orc_renormalize_null(C.byref(s), C.byref(m))
    h = 0.01
    for _ in range(1000):
        h = L.orc_adaptive_step(C.byref(s), C.byref(m), h, 1e-11, None)
        if s.x[1] < 0.5:
            break
    assert s.x[1] < 1.0


def test_legacy_coordinate_comparison(oracle):  # _legacy_src/integrator.rs:388-441
    import ctypes as C
    r, a, th = 3.0, 0.5, 1.57
    delta = r * r - 2.0 * r + a * a
    p_bl = [-1.0, 0.0, 0.0, 2.0]
    p_r_ks = p_bl[1] + (2.0 * r * 1.0 - a * 2.0) / delta
    L = oracle.lib()
    bl, ks = oracle.metric(oracle.KERR_BL, 1.0, a), oracle.metric(oracle.KERR_KS, 1.0, a)
    s_bl = oracle.make_state([0, r, th, 0] + p_bl)
    s_ks = oracle.make_state([0, r, th, 0, p_bl[0], p_r_ks, p_bl[2], p_bl[3]])
    assert abs(L.orc_hamiltonian(C.byref(s_bl), C.byref(bl)) -
               L.orc_hamiltonian(C.byref(s_ks), C.byref(ks))) < 1e-10


def test_short_input_is_echoed(oracle):  # gravitas-wasm/src/lib.rs:429-431
    out = oracle.integrate_ray_relativistic(1.0, 0.9, [1.0, 2.0, 3.0], 100, 1e-8, True)
    assert out.tolist() == [1.0, 2.0, 3.0]


def test_lut_single_row_quirk(oracle):  # spectrum.rs:82 with height = 1 -> g = 0.05
    lut = oracle.blackbody_lut(16, 1, 1e5).reshape(16, 4)
    assert np.all(lut[:, 3] == 1.0)
    full = oracle.blackbody_lut(16, 4, 1e5).reshape(4, 16, 4)
    assert np.array_equal(lut, full[0])


def test_planck_cutoff_and_cold_rows(oracle):  # spectrum.rs:14-16, 24-26
    assert oracle.lib().orc_planck_law(500e-9, 10.0) == 0.0
    xyz = np.zeros(3)
    oracle.lib().orc_integrate_planck_xyz(99.0, oracle._ptr(xyz))
    assert xyz.tolist() == [0.0, 0.0, 0.0]


@pytest.mark.parametrize("a", [0.0, 0.5, 0.9, 0.998])
def test_analytic_derivatives_match_finite_differences(oracle, a):
    """invariants/audit.rs:17-47 idea: dH/dr, dH/dtheta vs central differences of H."""
    import ctypes as C
    L = oracle.lib()
    for kind in (oracle.KERR_BL, oracle.KERR_KS):
        m = oracle.metric(kind, 1.0, a)
        p = [-1.0, 0.3, 1.1, 2.0]
        r, th, eps = 6.0, 1.1, 1e-6
        dr, dth = oracle.hamiltonian_derivatives(m, r, th, p)

        def H(rr, tt):
            s = oracle.make_state([0, rr, tt, 0] + p)
            return L.orc_hamiltonian(C.byref(s), C.byref(m))
        assert abs((H(r + eps, th) - H(r - eps, th)) / (2 * eps) - dr) < 1e-7
        assert abs((H(r, th + eps) - H(r, th - eps)) / (2 * eps) - dth) < 1e-7
