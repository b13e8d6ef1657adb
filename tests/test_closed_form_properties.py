"""Property tests of the closed forms next to the path (event horizon, photon sphere, ISCO,
time dilation; gravitas-core/src/metric/kerr.rs:75-123, gravitas-wasm/src/lib.rs:85-105).
The properties are the ones the reference's TypeScript suite states for the same formulas
(src/__tests__/physics/kerr-metric.test.ts:20-290 over src/physics/kerr-metric.ts), run here
against the oracle with hypothesis instead of fast-check; the GPU-marked test holds the engine's
FFI scalars to the oracle."""
import math

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

MASS = st.floats(0.1, 10.0)
SPIN = st.floats(-1.0, 1.0)


def _m(oracle, mass, spin):
    return oracle.metric(oracle.KERR_BL, mass, spin)


@settings(max_examples=100, deadline=None)
@given(MASS, SPIN)
def test_event_horizon_bounds(oracle, mass, spin):   # kerr-metric.test.ts:20-37
    rh = oracle.lib().orc_event_horizon(_m(oracle, mass, spin))
    assert 0.0 < rh <= 2.0 * mass + 1e-10 and rh >= mass - 1e-12


@settings(max_examples=50, deadline=None)
@given(MASS)
def test_zero_spin_radii(oracle, mass):              # :41-53, :108-120, :150-164
    L, m = oracle.lib(), _m(oracle, mass, 0.0)
    assert abs(L.orc_event_horizon(m) - 2.0 * mass) < 1e-8 * mass
    assert abs(L.orc_photon_sphere(m) - 3.0 * mass) < 1e-8 * mass
    assert abs(L.orc_isco(m, 0) - 6.0 * mass) < 1e-8 * mass and abs(L.orc_isco(m, 1) - 6.0 * mass) < 1e-8 * mass


@settings(max_examples=100, deadline=None)
@given(MASS, st.floats(0.0, 1.0), st.floats(0.0, 1.0))
def test_horizon_shrinks_with_spin(oracle, mass, s1, s2):   # :57-72
    lo, hi = sorted((s1, s2))
    L = oracle.lib()
    assert L.orc_event_horizon(_m(oracle, mass, hi)) <= L.orc_event_horizon(_m(oracle, mass, lo)) + 1e-10


@settings(max_examples=100, deadline=None)
@given(MASS, st.floats(0.0, 0.9))   # co-rotating convention: photon_sphere() keeps the sign of the spin
def test_photon_sphere_between_horizon_and_isco(oracle, mass, spin):   # :85-104 (spin 0 .. 0.9 there)
    L, m = oracle.lib(), _m(oracle, mass, spin)
    assert L.orc_event_horizon(m) < L.orc_photon_sphere(m) < L.orc_isco(m, 0) + 1e-9


@settings(max_examples=100, deadline=None)
@given(MASS, SPIN)
def test_isco_ordering(oracle, mass, spin):          # :133-146, :168-184
    L, m = oracle.lib(), _m(oracle, mass, abs(spin))
    pro, retro = L.orc_isco(m, 0), L.orc_isco(m, 1)
    assert pro <= retro + 1e-10 and pro >= L.orc_event_horizon(m) - 1e-9
    assert mass - 1e-9 <= pro <= 6.0 * mass + 1e-9 and 6.0 * mass - 1e-9 <= retro <= 9.0 * mass + 1e-9


@settings(max_examples=100, deadline=None)
@given(MASS, st.floats(1.0, 500.0), st.floats(1.0, 500.0))
def test_time_dilation_properties(oracle, mass, k1, k2):   # :198-290 (Schwarzschild, spin 0)
    L, m = oracle.lib(), oracle.metric(oracle.KERR_BL, mass, 0.0)
    r1, r2 = sorted((2.0 * mass * k1, 2.0 * mass * k2))
    f1, f2 = L.orc_time_dilation(m, r1, math.pi / 2), L.orc_time_dilation(m, r2, math.pi / 2)
    assert 0.0 <= f1 <= 1.0 and 0.0 <= f2 <= 1.0 and f2 >= f1 - 1e-15
    assert L.orc_time_dilation(m, 2.0 * mass * 0.999, math.pi / 2) == 0.0        # at / inside the horizon
    assert L.orc_time_dilation(m, 2.0 * mass * 5000.0, math.pi / 2) > 0.999
    # the FFI returns dt_coord / dt_proper and caps it at 100 inside (lib.rs:97-105)
    assert L.orc_compute_dilation(m, 2.0 * mass * 0.5) == 100.0
    assert abs(L.orc_compute_dilation(m, r2) * f2 - 1.0) < 1e-12 if f2 > 0 else True


@pytest.mark.gpu
def test_engine_closed_forms_match_oracle(engine_mod, oracle):
    rng = np.random.default_rng(5)
    L = oracle.lib()
    for mass, spin in zip(rng.uniform(0.1, 10.0, 12), rng.uniform(-1.2, 1.2, 12)):
        m = _m(oracle, float(mass), float(spin))
        with engine_mod.PhysicsEngine(float(mass), float(spin)) as e:
            assert abs(e.compute_horizon() - L.orc_event_horizon(m)) <= 1e-13 * mass
            assert abs(e.compute_isco() - L.orc_isco(m, 0)) <= 1e-12 * mass
            assert abs(e.compute_photon_sphere() - L.orc_photon_sphere(m)) <= 1e-12 * mass
            for r in (0.5 * mass, 2.5 * mass, 40.0 * mass):
                assert abs(e.compute_dilation(r) - L.orc_compute_dilation(m, r)) <= 1e-12 * L.orc_compute_dilation(m, r)
                a, b = e.compute_g_factor(r + 6 * mass, 1.5 * mass), L.orc_kerr_g_factor(r + 6 * mass, mass, spin, 1.5 * mass)
                assert abs(a - b) <= 1e-12 * max(1.0, abs(b))
