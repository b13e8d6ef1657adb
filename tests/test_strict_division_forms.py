"""The three division forms of the STRICT Kerr-Schild right-hand side (csrc/kerr_device.hpp: the
compiler's `/`, SharedDiv, SharedDivNoFixup) evaluated alone through grv_strict_rhs_probe.

The kernels pick a form per wave from operand guards and state that the choice is invisible in the
results.  Here every state is evaluated through each form: wherever a guard admits the state the six
derivatives must carry the bits of the plain-`/` form, and those must be the checker's
(orc_state_derivative, geodesic/hamiltonian.rs:13-35 over kerr.rs:412-499).  The states include the
exact cancellations the case analysis of SharedDivNoFixup is about -- Delta = +0 on a horizon radius,
r = M, Sigma = 2 r^2 -- which no live ray of a frame ever evaluates (they lie inside the horizon)."""
import numpy as np
import pytest


def _bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def _states(r, th, rng, hostile=False):
    n = r.size
    st = np.zeros((n, 8))
    st[:, 1] = r
    st[:, 2] = th
    st[:, 4] = -1.0
    st[:, 5] = rng.uniform(-2, 2, n)
    st[:, 6] = rng.uniform(-8, 8, n)
    st[:, 7] = rng.uniform(-8, 8, n)
    if hostile:
        k = n // 8
        st[:k, 4] = rng.choice([-1.0, 0.0, -0.0, 1e-310, 3.0, -1e200], k)     # p_t
        st[k:2 * k, 7] = rng.choice([0.0, -0.0, 1e-320, 1e250, -4.0], k)       # p_phi
        st[2 * k:3 * k, 5] = rng.choice([0.0, -0.0, 1e300, -1e-300, np.inf, np.nan], k)  # p_r
        st[3 * k:4 * k, 6] = rng.choice([0.0, -0.0, 1e300, np.nan], k)         # p_theta
    return st


def _differ(a, b):
    """element-wise: not bit-equal (NaNs equal NaNs: their payload and sign are not part of the contract)"""
    na, nb = np.isnan(a), np.isnan(b)
    return (na != nb) | (_bits(np.where(na, 0.0, a)) != _bits(np.where(nb, 0.0, b)))


def _same(a, b):
    return not _differ(np.asarray(a, np.float64), np.asarray(b, np.float64)).any()


def _oracle_rhs(oracle, m, state8):
    import ctypes as C
    s = oracle.make_state(state8)
    d = oracle.lib().orc_state_derivative(C.byref(s), C.byref(m))
    return np.array([d.x[0], d.x[1], d.x[2], d.x[3], d.p[1], d.p[2]])


def _check_forms(e, E, st):
    ieee = e.strict_rhs_probe(E.RHS_FORM_IEEE, st)
    shared = e.strict_rhs_probe(E.RHS_FORM_SHARED, st)
    nofix = e.strict_rhs_probe(E.RHS_FORM_NOFIXUP, st)
    assert np.all(ieee[:, 6] == E.RHS_FORM_IEEE)
    for name, got in (("shared", shared), ("nofixup", nofix)):
        bad = np.flatnonzero(_differ(got[:, :6], ieee[:, :6]).any(axis=1))
        assert bad.size == 0, (name, bad.size, st[bad[:3]], got[bad[:3]], ieee[bad[:3]])
    return ieee, shared, nofix


@pytest.mark.gpu
def test_forms_agree_on_random_and_hostile_states(engine_mod, oracle):
    bh = engine_mod
    E = bh.engine
    rng = np.random.default_rng(20261002)
    n = 400_000
    r = 10.0 ** rng.uniform(-1.5, 2.5, n)
    r[: n // 50] = rng.choice([0.0, -1.0, 1e-9, 3e7, np.inf], n // 50)            # outside every guard
    th = rng.uniform(0.0, np.pi, n)
    th[n // 50: n // 25] = rng.choice([0.0, np.pi / 2, np.pi, 1e-11, 1e-40, 3.0, -2.0, 40.0], n // 50)
    for mass, spin in ((1.0, 0.999), (1.0, 0.0), (2.5, -0.7), (1.0, 1.0)):
        st = _states(r, th, rng, hostile=True)
        with bh.PhysicsEngine(mass, spin) as e:
            ieee, shared, nofix = _check_forms(e, E, st)
        # the forms really ran where they were expected to: a = 0 never admits NOFIXUP, and with a
        # non-zero spin the bulk of the moderate states does
        frac_nf = (nofix[:, 6] == E.RHS_FORM_NOFIXUP).mean()
        frac_sh = (shared[:, 6] == E.RHS_FORM_SHARED).mean()
        assert frac_sh > 0.9
        assert (frac_nf == 0.0) if spin == 0.0 else (frac_nf > 0.8), (spin, frac_nf)
        # and the plain form is the checker's, on a sample (one ctypes call per state)
        m = oracle.metric(oracle.KERR_KS, mass, spin)
        for i in rng.integers(0, n, 3000):
            want = _oracle_rhs(oracle, m, st[i])
            assert _same(ieee[i, :6], want), (i, st[i], ieee[i, :6], want)


@pytest.mark.gpu
def test_forms_agree_on_exact_cancellations(engine_mod, oracle):
    """Delta = +0 (r on a horizon radius: M = 1.25, a = 0.75 -> r = 2.25 and 0.25, all exact in binary),
    dDelta/dr = +0 (r = M), Sigma - 2 r^2 = +0 (a^2 cos^2 = r^2, found by search with the specified
    sine).  On these the fixup-free sequence has a zero numerator; the case analysis says +0 comes out
    right on its own, the numerators negated after a cancellation are divided first, and
    Delta dSigma/dtheta keeps its fixup."""
    bh = engine_mod
    E = bh.engine
    rng = np.random.default_rng(5)
    th = rng.uniform(0.05, np.pi - 0.05, 3000)
    with bh.PhysicsEngine(1.25, 0.6) as e:  # a = 0.75
        for r0 in (2.25, 0.25, 1.25):
            st = _states(np.full(th.size, r0), th, rng)
            ieee, _, nofix = _check_forms(e, E, st)
            assert (nofix[:, 6] == E.RHS_FORM_NOFIXUP).all()
            if r0 != 1.25:
                # Delta = 0: g^rr = +0 -> dr has no p_r part; the checker agrees (sample)
                m = oracle.metric(oracle.KERR_KS, 1.25, 0.6)
                for i in range(0, th.size, 100):
                    assert _same(ieee[i, :6], _oracle_rhs(oracle, m, st[i]))
    # Sigma = 2 r^2 with M = a = 1: r = sqrt(a^2 cos^2) where the square of the rounded root is exact
    cand = rng.uniform(0.2, 1.3, 40000)
    s = oracle.ref_sin(cand)
    cos2 = 1.0 - np.maximum(s * s, 1e-12)
    r = np.sqrt(cos2)
    hit = (r * r == cos2) & ((r * r + cos2) - r * (2.0 * r) == 0.0)
    assert hit.sum() > 1000
    st = _states(r[hit], cand[hit], rng)
    with bh.PhysicsEngine(1.0, 1.0) as e:
        ieee, _, nofix = _check_forms(e, E, st)
    assert (nofix[:, 6] == E.RHS_FORM_NOFIXUP).all()
    m = oracle.metric(oracle.KERR_KS, 1.0, 1.0)
    for i in range(0, st.shape[0], 50):
        assert _same(ieee[i, :6], _oracle_rhs(oracle, m, st[i]))


def test_probe_entry_is_exported_and_validates(engine_mod):
    """CPU: the symbol exists and refuses bad arguments without touching a device."""
    L = engine_mod.load_library()
    assert L.grv_strict_rhs_probe(None, 0, 0, None, None) != 0
