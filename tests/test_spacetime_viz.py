"""Spacetime read-outs next to the path (gravitas-wasm/src/lib.rs:139-159,
214-306).  The oracle (oracle/viz_oracle.c) is pinned by the reference's own tests
(gravitas-core/src/spacetime/embedding.rs:113-129) and by closed forms; the GPU-marked
tests compare the engine's FFI (grid kernels + host scalars) with the oracle."""
import math

import numpy as np
import pytest

PI_2 = math.pi / 2


# ---- reference test pins + closed forms (CPU) -------------------------------------------
def test_flamm_at_horizon(oracle):  # embedding.rs:117-120
    assert oracle.lib().orc_flamm_height(2.0, 1.0) == 0.0


def test_flamm_at_large_r(oracle):  # embedding.rs:122-128
    z = oracle.lib().orc_flamm_height(100.0, 1.0)
    assert z > 0.0 and abs(z - 2.0 * math.sqrt(2.0 * 98.0)) < 0.1


def test_kretschner_schwarzschild_limit(oracle):  # curvature.rs:52-54: 48 M^2 / r^6
    L = oracle.lib()
    for r in (2.5, 6.0, 40.0):
        assert abs(L.orc_kretschner_kerr(r, 1.1, 1.0, 0.0) - 48.0 / r ** 6) <= 1e-15 * 48.0 / r ** 6 * 8
    assert L.orc_kretschner_kerr(0.0, PI_2, 1.0, 0.5) == math.inf  # ring singularity guard


def test_frame_drag_matches_equatorial_closed_form(oracle):  # kerr.rs:128-138 vs 143-152
    L = oracle.lib()
    m, s = 1.0, 0.9
    a = s * m
    for r in (2.0, 5.0, 30.0):
        closed = 2.0 * m * a / (r ** 3 + a * a * r + 2.0 * m * a * a)
        assert abs(L.orc_frame_dragging_omega(r, PI_2, m, s) - closed) < 1e-14
    assert L.orc_frame_dragging_omega(5.0, PI_2, 1.0, 0.0) == 0.0
    # spin is clamped into the metric (kerr.rs:48-54)
    assert L.orc_frame_dragging_omega(5.0, 1.0, 1.0, 1.7) == L.orc_frame_dragging_omega(5.0, 1.0, 1.0, 1.0)


def test_light_cone_tilt_limits(oracle):  # lightcone.rs:26-33
    L = oracle.lib()
    assert abs(L.orc_light_cone_tilt_bl(1e6, PI_2, 1.0, 0.5) - math.pi / 4) < 1e-5
    # inside the ergosphere (r < 2M on the equator) g_tt >= 0 -> pi/2
    assert L.orc_light_cone_tilt_bl(1.9, PI_2, 1.0, 0.9) == PI_2
    t = [L.orc_light_cone_tilt_bl(r, PI_2, 1.0, 0.0) for r in (2.1, 3.0, 10.0, 100.0)]
    assert all(b > a for a, b in zip(t, t[1:]))  # cones open up away from the hole


def test_ergosphere_radius(oracle):  # kerr.rs:157-167
    L = oracle.lib()
    assert abs(L.orc_ergosphere_radius(PI_2, 1.0, 0.9) - 2.0) < 1e-15
    rp = 1.0 + math.sqrt(1.0 - 0.81)
    assert abs(L.orc_ergosphere_radius(0.0, 1.0, 0.9) - rp) < 1e-15


def test_proper_distance_schwarzschild(oracle):  # embedding.rs:51-65 vs the closed form
    L = oracle.lib()

    def exact(r):  # integral of (1 - 2/r)^-1/2 dr, M = 1
        return math.sqrt(r * (r - 2.0)) + 2.0 * math.log(math.sqrt(r) + math.sqrt(r - 2.0))
    got = L.orc_proper_distance(4.0, 20.0, 2000, 1.0, 0.0)
    assert abs(got - (exact(20.0) - exact(4.0))) < 1e-5
    assert got > 16.0
    assert L.orc_proper_distance(20.0, 4.0, 2000, 1.0, 0.0) == got  # argument order (embedding.rs:53)


def test_field_and_mesh_layouts(oracle):
    f = oracle.scalar_field(0, 1.0, 0.9, 2.0, 10.0, 5, 7).reshape(5, 7, 3)
    assert np.allclose(f[:, 0, 0], np.linspace(2.0, 10.0, 5))
    assert np.allclose(f[0, :, 1], 0.1 + (math.pi - 0.2) * np.arange(7) / 6, atol=1e-6)
    assert abs(f[2, 3, 2] - oracle.lib().orc_kretschner_kerr(6.0, PI_2, 1.0, 0.9)) < 1e-6
    mesh = oracle.embedding_mesh(1.0, 0.0, 2.0, 20.0, 4, 8).reshape(4, 8, 3)
    assert np.allclose(np.hypot(mesh[:, :, 0], mesh[:, :, 2]), np.linspace(2.0, 20.0, 4)[:, None], atol=1e-5)
    assert mesh[0, 0, 1] == 0.0 and np.all(mesh[1:, :, 1] < 0.0)  # funnel points down
    kerr = oracle.embedding_mesh(1.0, 0.9, 3.0, 20.0, 4, 8).reshape(4, 8, 3)
    assert kerr[-1, 0, 1] == 0.0 and np.all(np.diff(kerr[:, 0, 1]) > 0)  # height measured from r_max
    ergo = oracle.ergosphere_mesh(1.0, 0.9, 9, 6).reshape(9, 6, 3)
    assert np.allclose(np.linalg.norm(ergo[4], axis=1), 2.0, atol=1e-6)


# ---- engine vs oracle (GPU box) ---------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("spin", [0.0, 0.9, 1.3])
def test_engine_spacetime_readouts_match_oracle(engine_mod, oracle, spin):
    L = oracle.lib()
    with engine_mod.PhysicsEngine(1.0, spin) as e:
        for r, th in ((2.5, 0.4), (6.0, PI_2), (30.0, 2.9)):
            for got, want in ((e.compute_kretschner(r, th), L.orc_kretschner_kerr(r, th, 1.0, spin)),
                              (e.compute_light_cone_tilt(r, th), L.orc_light_cone_tilt_bl(r, th, 1.0, spin)),
                              (e.compute_frame_drag_omega(r, th), L.orc_frame_dragging_omega(r, th, 1.0, spin))):
                assert got == want or (math.isinf(got) and math.isinf(want))   # specified sin/cos/atan: same bits
        assert e.compute_flamm_height(2.0) == 0.0
        assert e.compute_flamm_height(100.0) == L.orc_flamm_height(100.0, 1.0)
        a, b = e.compute_proper_distance(20.0, 4.0, 500), L.orc_proper_distance(20.0, 4.0, 500, 1.0, spin)
        assert a == b
        for kind, fn in ((0, e.generate_curvature_field), (1, e.generate_tilt_field),
                         (2, e.generate_frame_drag_field)):
            got = fn(2.2, 40.0, 33, 17)                      # one thread per grid point
            want = oracle.scalar_field(kind, 1.0, spin, 2.2, 40.0, 33, 17)
            assert got.shape == want.shape
            assert np.array_equal(got, want)
        got = e.generate_embedding_mesh(2.5, 30.0, 24, 16)
        want = oracle.embedding_mesh(1.0, spin, 2.5, 30.0, 24, 16)
        assert np.array_equal(got, want)
        got = e.generate_ergosphere_mesh(17, 12)
        want = oracle.ergosphere_mesh(1.0, spin, 17, 12)
        assert np.array_equal(got, want)
        # empty grids are empty loops upstream
        assert e.generate_curvature_field(2.0, 10.0, 0, 5).size == 0


@pytest.mark.gpu
def test_engine_disk_lut_ptr_tracks_last_generate(engine_mod):  # lib.rs:107-114
    with engine_mod.PhysicsEngine(1.0, 0.5) as e:
        view = e.get_disk_lut_view()
        assert view.shape == (512,) and not view.any()
        lut = e.generate_disk_lut()
        assert np.array_equal(e.get_disk_lut_view(), lut) and lut.max() == 1.0
