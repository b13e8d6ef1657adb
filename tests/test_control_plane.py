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


# ---- 8f-1 in the shading: the Page-Thorne table as the disk's radial temperature profile ------
def test_page_thorne_profile_reader(oracle):
    """orc_disk_lut_profile reads the 512-entry table of generate_temperature_lut
    (physics/disk.rs:175-201) the way its LINEAR / CLAMP_TO_EDGE texture is read: entry i sits at
    r = rin + i/511 (rout - rin), halfway between two entries is their mean, both ends clamp; and
    the profile it yields keeps disk.rs:284-308 (peak within 3 r_isco, positive)."""
    import ctypes as C
    L = oracle.lib()
    L.orc_disk_lut_profile.restype = C.c_double
    L.orc_disk_lut_profile.argtypes = [C.c_void_p, C.c_uint32, C.c_double, C.c_double, C.c_double]
    for spin in (0.0, 0.9, 0.999):
        lut = oracle.temperature_lut(1.0, spin)
        rin = L.orc_isco(oracle.metric(oracle.KERR_BL, 1.0, spin), 0)
        rout = 50.0
        ptr = lut.ctypes.data_as(C.c_void_p)
        for i in (0, 1, 17, 255, 510, 511):
            r = rin + i / 511.0 * (rout - rin)
            assert abs(L.orc_disk_lut_profile(ptr, 512, r, rin, rout) - float(lut[i])) < 1e-6
        for i in (0, 100, 300):
            r = rin + (i + 0.5) / 511.0 * (rout - rin)
            want = 0.5 * (float(lut[i]) + float(lut[i + 1]))
            assert abs(L.orc_disk_lut_profile(ptr, 512, r, rin, rout) - want) < 1e-6
        assert L.orc_disk_lut_profile(ptr, 512, rin - 3.0, rin, rout) == float(lut[0]) == 0.0
        assert L.orc_disk_lut_profile(ptr, 512, 80.0, rin, rout) == float(lut[511])
        rs = np.linspace(rin, rout, 2000)
        prof = np.array([L.orc_disk_lut_profile(ptr, 512, float(r), rin, rout) for r in rs])
        assert prof.max() > 0.99 and prof.min() >= 0.0
        if spin == 0.0:   # disk.rs:284-308 is stated for Schwarzschild
            assert rs[prof.argmax()] < 3.0 * rin


def test_page_thorne_profile_changes_the_shaded_frame(oracle):
    W, H = 64, 36
    th = np.deg2rad(80.0)
    cam = oracle.camera_look_at((60 * np.sin(th), 60 * np.cos(th), 0.0), aspect=W / H)
    a = oracle.render_frame(cam, oracle.frame_params(W, H, spin=0.9), None, nthreads=4)
    b = oracle.render_frame(cam, oracle.frame_params(W, H, spin=0.9, disk_profile=1), None, nthreads=4)
    assert np.array_equal(a["states"], b["states"]) and np.array_equal(a["steps"], b["steps"])
    lit = a["rgba"][..., :3].sum(-1) > 0
    assert lit.any() and np.array_equal(lit, b["rgba"][..., :3].sum(-1) > 0)     # same disk pixels
    assert not np.array_equal(a["rgba"], b["rgba"])
    # the table is normalised to its peak, the closed form peaks at 0.49: hotter -> brighter overall
    assert b["rgba"][..., :3].sum() > a["rgba"][..., :3].sum()


@pytest.mark.gpu
@pytest.mark.parametrize("spin,theta_deg", [(0.999, 97.0), (0.5, 75.0), (0.0, 60.0)])
def test_page_thorne_shaded_frame_is_bit_exact(engine_mod, oracle, spin, theta_deg):
    """GrvRenderParams.disk_profile = PAGE_THORNE: finalize_frame_kernel reads the device table
    (staged in LDS beside the Planck rows); pixels equal the oracle's bit for bit (STRICT) and
    to 1e-5 of the peak (FAST)."""
    import torch
    bh = engine_mod
    W, H = 160, 90
    th = np.deg2rad(theta_deg)
    eye = (60 * np.sin(th), 60 * np.cos(th), 0.0)
    ref = oracle.render_frame(oracle.camera_look_at(eye, aspect=W / H),
                              oracle.frame_params(W, H, spin=spin, disk_profile=1), None, nthreads=8)
    peak = float(ref["rgba"][..., :3].max())
    assert peak > 0
    with bh.PhysicsEngine(1.0, spin) as e:
        cam = bh.camera_look_at(eye, aspect=W / H)
        for arith in (bh.ARITH_STRICT, bh.ARITH_FAST):
            p = bh.render_params(W, H, arith=arith, disk_profile=bh.DISK_PROFILE_PAGE_THORNE)
            rgba = torch.zeros(W * H, 4, dtype=torch.float32, device="cuda:0")
            e.render_frame_device(cam, p, rgba=rgba)
            torch.cuda.synchronize()
            got = rgba.cpu().numpy().reshape(H, W, 4)
            if arith == bh.ARITH_STRICT:
                assert np.array_equal(got, ref["rgba"])
            else:
                assert np.abs(got - ref["rgba"]).max() <= 1e-5 * peak
        # the table follows update_params (it is keyed by mass and spin)
        e.update_params(1.0, 0.3)
        ref2 = oracle.render_frame(oracle.camera_look_at(eye, aspect=W / H),
                                   oracle.frame_params(W, H, spin=0.3, disk_profile=1), None, nthreads=8)
        p = bh.render_params(W, H, arith=bh.ARITH_STRICT, disk_profile=bh.DISK_PROFILE_PAGE_THORNE)
        e.render_frame_device(cam, p, rgba=rgba)
        torch.cuda.synchronize()
        assert np.array_equal(rgba.cpu().numpy().reshape(H, W, 4), ref2["rgba"])
