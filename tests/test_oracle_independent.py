"""The oracle against a SECOND implementation that shares nothing with it: tests/independent_reference.py is
plain Python written from the Rust sources (integrator.rs:53-190, mod.rs:180-265, kerr.rs:266-499,
renormalization.rs:13-45, invariants/mod.rs:25-37) with the host libm's sin / cos / pow.  On the committed
ray list tests/golden/rays_v2.npz -- 542 rays: {Kerr-Schild, Boyer-Lindquist} x tolerance {1e-6, 1e-8, 1e-10}
x a in {0, 0.5, 0.9, 0.999}, plus rays on the controller's forced-minimum-step branch -- both must agree on
the termination class and on steps_taken of every ray, and on the end state to 1e-6.

The reference holds no executed integrate() result, and its toolchain is absent here; this test is
what stands between the oracle and a transcription error in the accept / reject / forced-step logic
(integrator.rs:72-107).  tests/test_ref_vectors.py consumes the same file, so the day cargo is available
tools/ref_vectors pins all three to the reference itself.

Boyer-Lindquist rays that END at the horizon are compared by class and step count only: the chart is singular
there (dt / dlambda, dphi / dlambda ~ 1 / Delta), a different last bit in sin / cos moves t and phi by O(1) --
the same carve-out tests/test_ref_vectors.py applies.  Kerr-Schild rays are compared everywhere."""
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import independent_reference as ind  # noqa: E402

GOLD = os.path.join(HERE, "golden", "rays_v2.npz")
TOL_END = 1e-6


def _run_case(args):
    kind, spin, tol, max_steps, esc, renorm, h0, rays = args
    hole = ind.KerrHole(1.0, spin, ind.KS if kind == 1 else ind.BL)
    out = []
    for y0 in rays:
        c = {"tries": 0, "forced": 0}
        y, term, steps, drift = ind.integrate(y0, hole, tolerance=tol, initial_step=h0, max_steps=max_steps,
                                              escape_radius=esc, renormalize_interval=renorm, counters=c)
        out.append((y, term, steps, drift, c["tries"], c["forced"]))
    return out


@pytest.fixture(scope="module")
def results():
    z = np.load(GOLD)
    keys = [str(c) for c in z["cases"]]
    jobs = []
    for k in keys:
        kind, spin, method, tol, max_steps, step, esc, renorm, h0 = z[k + "_meta"]
        assert int(method) == 0
        jobs.append((int(kind), float(spin), float(tol), int(max_steps), float(esc), int(renorm), float(h0),
                     [tuple(r) for r in z[k + "_in"]]))
    with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        res = list(ex.map(_run_case, jobs))
    return z, dict(zip(keys, res))


def test_ray_list_covers_what_it_claims():
    z = np.load(GOLD)
    keys = [str(c) for c in z["cases"]]
    n = sum(z[k + "_in"].shape[0] for k in keys)
    assert n >= 500
    combos = {(int(z[k + "_meta"][0]), float(z[k + "_meta"][1]), float(z[k + "_meta"][3])) for k in keys}
    for kind in (0, 1):
        for spin in (0.0, 0.5, 0.9, 0.999):
            for tol in (1e-6, 1e-8, 1e-10):
                assert (kind, spin, tol) in combos
    terms = np.concatenate([z[k + "_term"] for k in keys])
    assert (terms == 1).sum() >= 100 and (terms == 2).sum() >= 300 and (terms == 3).sum() >= 4


def test_oracle_agrees_with_the_independent_implementation(results):
    z, res = results
    worst, worst_key, n_cmp, forced_rays, forced_steps = 0.0, None, 0, 0, 0
    for k, rows in res.items():
        kind = int(z[k + "_meta"][0])
        for i, (y, term, steps, drift, tries, forced) in enumerate(rows):
            assert term == int(z[k + "_term"][i]), (k, i, "class", term, int(z[k + "_term"][i]))
            assert steps == int(z[k + "_steps"][i]), (k, i, "steps", steps, int(z[k + "_steps"][i]))
            forced_rays += forced > 0
            forced_steps += forced
            if kind == 0 and term == ind.HORIZON:
                continue  # Boyer-Lindquist at the horizon: class and step count only (module docstring)
            want = z[k + "_out"][i]
            got = np.array(y)
            if not np.all(np.isfinite(want)):
                assert np.array_equal(np.isfinite(got), np.isfinite(want)), (k, i)
                continue
            err = float((np.abs(got - want) / np.maximum(1.0, np.abs(want))).max())
            n_cmp += 1
            if err > worst:
                worst, worst_key = err, (k, i)
            assert err <= TOL_END, (k, i, err)
            d = float(z[k + "_drift"][i])
            assert abs(drift - d) <= 1e-6 * max(1.0, d) + 1e-9, (k, i, drift, d)
    print("independent implementation vs oracle: %d end states compared, worst relative difference %.3g at %s; "
          "%d rays took %d forced minimum steps" % (n_cmp, worst, worst_key, forced_rays, forced_steps))
    assert n_cmp >= 400
    # the forced-minimum-step branch (integrator.rs:99-104) is exercised: eight rays under a tolerance no step meets
    assert forced_rays >= 8 and forced_steps >= 8 * 55


def test_forced_step_cases_walk_the_forced_branch_only(results):
    z, res = results
    for k in ("ks_a0.9_forced", "bl_a0.9_forced"):
        for (y, term, steps, drift, tries, forced) in res[k]:
            # (a try whose error estimate is exactly 0 is accepted as it stands, integrator.rs:78-82: not every one of
            # the 60 steps has to be a forced one)
            assert term == ind.MAXSTEPS and steps == 60 and forced >= 55
