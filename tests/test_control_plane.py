"""Disk / shadow / SAB rows: the oracle is pinned by the
reference's own tests (gravitas-core/src/physics/disk.rs:226-309, shadow.rs:260-335);
the GPU-marked tests compare the engine's FFI with the oracle."""
import math

import numpy as np
import pytest

PI_2 = math.pi / 2


# ---- reference test pins (CPU) ---------------------------------------------------------
def test_flux_zero_at_isco(oracle):  # disk.rs:226-236
    L = oracle.lib()
    isco = L.orc_isco(oracle.metric(oracle.KERR_BL, 1.0, 0.0), 0)
    assert abs(L.orc_page_thorne_flux(isco, 1.0, 0.0, 1.0)) < 1e-10


def test_flux_positive_outside_isco(oracle):  # disk.rs:238-244
    assert oracle.lib().orc_page_thorne_flux(9.0, 1.0, 0.0, 1.0) > 0.0


def test_flux_decays_at_large_r(oracle):  # disk.rs:246-257
    L = oracle.lib()
    assert L.orc_page_thorne_flux(10.0, 1.0, 0.0, 1.0) > L.orc_page_thorne_flux(40.0, 1.0, 0.0, 1.0)


def test_spinning_bh_has_higher_peak_flux(oracle):  # disk.rs:259-282
    L = oracle.lib()
    i0 = L.orc_isco(oracle.metric(oracle.KERR_BL, 1.0, 0.0), 0)
    i9 = L.orc_isco(oracle.metric(oracle.KERR_BL, 1.0, 0.9), 0)
    m0 = max(L.orc_page_thorne_flux(i0 + i * 0.2, 1.0, 0.0, 1.0) for i in range(1, 100))
    m9 = max(L.orc_page_thorne_flux(i9 + i * 0.2, 1.0, 0.9, 1.0) for i in range(1, 100))
    assert m9 > m0


def test_temperature_profile_peaks_near_isco(oracle):  # disk.rs:284-308
    L = oracle.lib()
    isco = 6.0
    r = np.array([isco + i / 199 * (50.0 - isco) for i in range(200)])
    t = np.array([L.orc_disk_temperature(x, 1.0, 0.0, 1.0) for x in r])
    assert r[t.argmax()] < 3 * isco and t.max() > 0
    lut = oracle.temperature_lut(1.0, 0.0)
    assert lut.max() == 1.0 and lut[0] == 0.0 and lut.shape == (512,)


def test_schwarzschild_shadow_is_circle(oracle):  # shadow.rs:260-276
    pts = oracle.bardeen_shadow(1.0, 0.0, PI_2, 100)
    assert pts.shape == (100, 2)
    assert np.all(np.abs(np.hypot(pts[:, 0], pts[:, 1]) - 3 * math.sqrt(3)) < 0.2)


def test_kerr_shadow_is_asymmetric(oracle):  # shadow.rs:278-296
    pts = oracle.bardeen_shadow(1.0, 0.9, PI_2, 100)
    assert pts.shape == (200, 2)
    assert abs(abs(pts[:, 0].min()) - abs(pts[:, 0].max())) > 0.1


def test_shadow_shrinks_with_spin(oracle):  # shadow.rs:298-324
    r0 = np.hypot(*oracle.bardeen_shadow(1.0, 0.0, PI_2, 100).T).mean()
    r9 = np.hypot(*oracle.bardeen_shadow(1.0, 0.9, PI_2, 100).T).mean()
    assert r9 < r0


def test_shadow_edges_match_equatorial_critical_impact_parameters(oracle):
    """Independent check: at theta_obs = pi/2 the curve's alpha extremes are the Bardeen
    prograde / retrograde critical impact parameters (same numbers as test_oracle_physics)."""
    a = 0.9
    pts = oracle.bardeen_shadow(1.0, a, PI_2, 400)
    b_pro = 6.0 * math.cos(math.acos(-a) / 3.0) - a
    b_ret = 6.0 * math.cos(math.acos(a) / 3.0) + a
    # alpha = a sin(theta) - xi / sin(theta): the reference adds a sin(theta) (shadow.rs:158)
    assert abs((pts[:, 0].max() - a) - b_ret) < 2e-2
    assert abs((pts[:, 0].min() - a) + b_pro) < 2e-2


def test_tick_sab_layout_and_overflow_quirk(oracle):  # lib.rs:308-409, SURVEY F10
    e = oracle.sab_engine(1.0, 0.9)
    sab = oracle.tick_sab(e, 0.016)
    assert sab[64:67].tolist() == [0.0, 0.0, 20.0]            # camera position (camera.rs:28)
    assert sab[72:76].tolist() == [0.0, 1.0, 0.0, 0.0]        # orientation xyzw
    assert abs(sab[128] - 1.4358899) < 1e-6 and abs(sab[130] - 1.0) < 1e-7 and abs(sab[131] - 0.9) < 1e-7
    assert sab[143] == 64.0                                   # point count at PHYSICS+15
    # 64 points x 2 floats from PHYSICS+16 = indices 144..271: the last 8 points overrun
    # TELEMETRY (256); the sequence float at [256] is alpha_56 + 1
    curve = oracle.bardeen_shadow(1.0, 0.9, math.acos(0.0), 32)
    assert np.allclose(sab[144:272:2][:56], curve[:56, 0].astype(np.float32))
    assert sab[256] == np.float32(np.float32(curve[56, 0]) + np.float32(1.0))
    sab2 = oracle.tick_sab(e, 0.016)
    assert sab2[256] == sab[256]                              # cleared and rewritten every tick


def test_tick_sab_inputs_consumed_and_camera_moves(oracle):
    e = oracle.sab_engine(1.0, 0.5)
    e.sab[1], e.sab[3] = 10.0, 2.0     # mouse_dx, zoom_delta
    sab = oracle.tick_sab(e, 0.02)
    assert sab[1] == 0.0 and sab[2] == 0.0 and sab[3] == 0.0
    yaw = -10.0 * 2.0 * 0.02
    z = 20.0 * (1.0 + 2.0 * 0.02)
    assert np.allclose(sab[64:67], [z * math.sin(yaw), 0.0, z * math.cos(yaw)], atol=1e-5)


# ---- engine vs oracle (GPU box) --------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("spin", [0.0, 0.9, 0.998])
def test_engine_disk_and_shadow_match_oracle(engine_mod, oracle, spin):
    with engine_mod.PhysicsEngine(1.0, spin) as e:
        lut = e.generate_disk_lut()          # GPU kernel
        ref = oracle.temperature_lut(1.0, spin)
        assert np.array_equal(lut, ref)   # Page-Thorne flux with the specified pow: same bits
        for r in (3.0, 7.0, 25.0):
            a, b = e.compute_disk_flux(r), oracle.lib().orc_page_thorne_flux(r, 1.0, spin, 1.0)
            assert a == b
        for th in (PI_2, 1.0, 1e-12):
            got = e.compute_shadow_curve(th, 32).reshape(-1, 2)
            want = oracle.bardeen_shadow(1.0, spin, th, 32)
            assert got.shape == want.shape
            assert np.array_equal(got, want.astype(np.float32))   # host code, host libm on both sides
        assert e.compute_shadow_radius() == 3.0 * math.sqrt(3.0)


@pytest.mark.gpu
def test_engine_tick_sab_matches_oracle(engine_mod, oracle):
    with engine_mod.PhysicsEngine(1.0, 0.9) as e:
        o = oracle.sab_engine(1.0, 0.9)
        e.set_auto_spin(True)
        o.camera.auto_spin = 1
        e.set_camera_state(3.0, 4.0, 12.0)
        o.camera.position[0], o.camera.position[1], o.camera.position[2] = 3.0, 4.0, 12.0
        view = e.sab_view()
        for k in range(3):
            view[1], view[3] = 0.5 * k, -0.1
            o.sab[1], o.sab[3] = 0.5 * k, -0.1
            e.tick_sab(0.016)
            want = oracle.tick_sab(o, 0.016)
            assert np.array_equal(view, np.asarray(want, np.float32))
        # attach_sab redirects the tick to caller memory (lib.rs:74, 309-313)
        ext = np.zeros(2048, np.float32)
        e.attach_sab(ext)
        e.tick_sab(0.016)
        assert ext[128] == view[128] and ext[143] == 64.0
