"""Trajectory.path (geodesic/mod.rs:150-161): grv_integrate_paths against the oracle's
orc_integrate_path.  STRICT arithmetic: every recorded point, the counts and the end states are
the oracle's bits, including point 0 = the state as handed in, pushed BEFORE the initial
renormalize_null (mod.rs:193-197 vs :200).  FAST: same step counts, points within the FAST
contract's tolerance (1e-5 relative, tests/test_gpu_parity.py) and the end state bitwise equal to
the plain batch call's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bh(engine_mod):
    return engine_mod


def _rays(n, seed):
    rng = np.random.default_rng(seed)
    st = np.zeros((n, 8))
    st[:, 1] = rng.uniform(8.0, 40.0, n)                 # r
    st[:, 2] = rng.uniform(0.3, np.pi - 0.3, n)          # theta
    st[:, 3] = rng.uniform(-np.pi, np.pi, n)             # phi
    st[:, 4] = -1.0                                      # p_t
    st[:, 5] = rng.choice([-1.0, 1.0], n) * rng.uniform(0.2, 1.0, n)
    st[:, 6] = rng.uniform(-3.0, 3.0, n)                 # p_theta
    st[:, 7] = rng.uniform(-6.0, 6.0, n)                 # p_phi
    st[0] = [0.0, 20.0, np.pi / 2, 0.0, -1.0, -1.0, 0.0, 3.5]  # the doc-test ray, mod.rs:169-179
    return st


CASES = [  # (metric kind, method, step_size, max_steps, spin)
    ("KERR_KS", 0, 0.0, 600, 0.9),
    ("KERR_BL", 0, 0.0, 600, 0.9),
    ("SCHWARZSCHILD", 0, 0.0, 400, 0.0),
    ("KERR_KS", 1, 0.25, 300, 0.999),   # RK4
    ("KERR_KS", 2, 0.05, 300, 0.5),     # symplectic
]


@pytest.mark.parametrize("kind,method,step,max_steps,spin", CASES)
def test_paths_are_the_oracles_bits_under_strict(bh, oracle, kind, method, step, max_steps, spin):
    po = oracle
    st = _rays(48, 11)
    kind_id = getattr(bh, kind)
    with bh.PhysicsEngine(1.0, spin) as e:
        o = bh.engine.default_options(method=method, metric_kind=kind_id, max_steps=max_steps, step_size=step,
                                      arith=bh.ARITH_STRICT, record_path=1)
        got = e.integrate_paths(st, o)
        plain = e.integrate_batch(st, o)
    m = po.metric(kind_id, 1.0, spin)
    oo = po.options(method=method, max_steps=max_steps, step_size=step)
    for i in range(st.shape[0]):
        t, path = po.integrate_path(st[i], m, oo, cap=max_steps + 1)
        assert int(got["counts"][i]) == path.shape[0] == int(t.steps_taken) + 1, i
        assert np.array_equal(got["paths"][i], path, equal_nan=True), (i, kind, method)
        # point 0 is the caller's state, not the renormalised one
        assert np.array_equal(got["paths"][i][0], st[i])
        assert int(got["steps"][i]) == int(t.steps_taken) and int(got["term"][i]) == int(t.termination)
        if path.shape[0] > 1:
            assert np.array_equal(got["paths"][i][-1], got["states"][i], equal_nan=True)
    # recording changes nothing else
    for k in ("states", "steps", "term", "drift"):
        assert np.array_equal(got[k], plain[k], equal_nan=True), k


def test_paths_truncate_at_max_points_and_none_without_record_path(bh, oracle):
    po = oracle
    st = _rays(20, 5)
    with bh.PhysicsEngine(1.0, 0.9) as e:
        o = bh.engine.default_options(max_steps=500, record_path=1)
        full = e.integrate_paths(st, o)
        short = e.integrate_paths(st, o, max_points=17)
        one = e.integrate_paths(st, o, max_points=1)
        zero = e.integrate_paths(st, o, max_points=0)
        o0 = bh.engine.default_options(max_steps=500, record_path=0)
        none = e.integrate_paths(st, o0)
        # max_steps = 0: the loop body never runs, the Vec holds the initial state alone (mod.rs:193-203)
        z = e.integrate_paths(st, bh.engine.default_options(max_steps=0, record_path=1))
    assert np.array_equal(short["counts"], full["counts"]) and np.array_equal(one["counts"], full["counts"])
    assert np.array_equal(zero["counts"], full["counts"])
    for i in range(st.shape[0]):
        k = min(17, int(full["counts"][i]))
        assert short["paths"][i].shape == (k, 8) and np.array_equal(short["paths"][i], full["paths"][i][:k])
        assert np.array_equal(one["paths"][i], st[i:i + 1])
        assert zero["paths"][i].shape == (0, 8)
    assert none["paths"] is None and not none["counts"].any()
    assert np.array_equal(none["states"], full["states"])
    assert (z["counts"] == 1).all() and all(np.array_equal(z["paths"][i], st[i:i + 1]) for i in range(20))
    assert (z["term"] == 3).all() and (z["steps"] == 0).all()


def test_paths_fast_contract(bh, oracle):
    po = oracle
    st = _rays(64, 3)
    with bh.PhysicsEngine(1.0, 0.999) as e:
        o = bh.engine.default_options(max_steps=800, arith=bh.ARITH_FAST, record_path=1)
        got = e.integrate_paths(st, o)
        plain = e.integrate_batch(st, o)
    m = po.metric(bh.KERR_KS, 1.0, 0.999)
    oo = po.options(max_steps=800)
    worst = 0.0
    for i in range(st.shape[0]):
        t, path = po.integrate_path(st[i], m, oo, cap=801)
        if int(got["counts"][i]) != path.shape[0]:
            continue  # an accept test on the other side of a rounding (counted in test_gpu_parity)
        d = np.abs(got["paths"][i] - path) / np.maximum(1.0, np.abs(path))
        worst = max(worst, float(d.max()))
    assert worst <= 1e-5, worst
    assert np.array_equal(got["steps"], plain["steps"]) and np.array_equal(got["term"], plain["term"])
    assert np.array_equal(got["states"], plain["states"], equal_nan=True)


def test_paths_argument_errors(bh):
    st = _rays(4, 1)
    with bh.PhysicsEngine(1.0, 0.5) as e:
        with pytest.raises(bh.GravitasError):
            e.integrate_paths(st, bh.engine.default_options(record_path=1, reserved=7))
        with pytest.raises(bh.GravitasError):
            e.integrate_paths(st, bh.engine.default_options(record_path=1, method=9))


def test_golden_paths(bh):
    """The committed Trajectory.path fixtures (tests/golden/paths_v1.npz), STRICT: bit for bit."""
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z, pz = np.load(os.path.join(gold, "rays_v1.npz")), np.load(os.path.join(gold, "paths_v1.npz"))
    for key in [str(c) for c in pz["cases"]]:
        kind, spin, method, tol, max_steps, step, esc, renorm, h0 = z[key + "_meta"]
        with bh.PhysicsEngine(1.0, float(spin)) as e:
            o = bh.engine.default_options(method=int(method), metric_kind=int(kind), tolerance=float(tol),
                                          initial_step=float(h0), max_steps=int(max_steps), escape_radius=float(esc),
                                          renormalize_interval=int(renorm), step_size=float(step), record_path=1)
            got = e.integrate_paths(z[key + "_in"][:6], o)
        assert np.array_equal(got["counts"], pz[key + "_counts"]), key
        for i in range(6):
            n = int(pz[key + "_counts"][i])
            assert np.array_equal(got["paths"][i], pz[key + "_paths"][i, :n], equal_nan=True), (key, i)
