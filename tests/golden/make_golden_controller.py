"""Generates tests/golden/rays_v2.npz: the ray list that pins the RKF45 controller three ways --
the oracle (oracle/gravitas_oracle.c, whose outputs are stored here), the independent Python
implementation (tests/independent_reference.py, tests/test_oracle_independent.py) and, on a machine
with cargo, the reference itself (tools/ref_vectors -> tests/golden/ref_rays_v2.json,
tests/test_ref_vectors.py).  Same layout as rays_v1.npz (make_golden.py), AdaptiveRKF45 only:

  {Kerr-Schild, Boyer-Lindquist} x tolerance {1e-6, 1e-8, 1e-10} x a in {0, 0.5, 0.9, 0.999}: 24 cases x 22 rays
  + two cases that live on the forced-minimum-step branch (integrator.rs:99-104): a tolerance no step can
    meet (all but the exact-zero-error steps are forced ones), in both charts;
  + Boyer-Lindquist plunges at a = 0.999 that run into the chart's coordinate singularity at the horizon
    (long runs of rejected tries).

    python tests/golden/make_golden_controller.py      # rewrites tests/golden/rays_v2.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402


def camera_fan(seed, n):
    """rays as a pinhole camera would send them: from (r0, theta0) towards the hole with impact parameters on
    both sides of the critical one, so a case holds captured and escaping rays and some that circle first"""
    rng = np.random.default_rng(seed)
    st = np.zeros((n, 8))
    st[:, 1] = rng.uniform(6.0, 60.0, n)
    st[:, 2] = rng.uniform(0.25, np.pi - 0.25, n)
    st[:, 3] = rng.uniform(0.0, 2 * np.pi, n)
    st[:, 4] = -1.0
    st[:, 5] = -rng.uniform(0.6, 1.0, n)
    b = rng.uniform(-9.0, 9.0, n)          # p_phi = impact parameter
    q = rng.uniform(-6.0, 6.0, n)          # p_theta
    st[:, 6] = q
    st[:, 7] = b
    return st


def add(out, cases, key, kind, spin, opt, st):
    res = po.integrate_batch(po.metric(kind, 1.0, spin), opt, st, nthreads=4)
    cases.append(key)
    out[key + "_in"] = st
    out[key + "_out"] = res["states"]
    out[key + "_steps"] = res["steps"]
    out[key + "_term"] = res["term"]
    out[key + "_drift"] = res["drift"]
    out[key + "_meta"] = np.array([kind, spin, po.METHOD_RKF45, opt.tolerance, opt.max_steps, opt.step_size,
                                   opt.escape_radius, opt.renormalize_interval, opt.initial_step])
    return res


def main():
    out, cases = {}, []
    seed = 7000
    for kind, kname in ((po.KERR_KS, "ks"), (po.KERR_BL, "bl")):
        for spin in (0.0, 0.5, 0.9, 0.999):
            for tol, tname in ((1e-6, "t6"), (1e-8, "t8"), (1e-10, "t10")):
                seed += 1
                opt = po.options(max_steps=6000, tolerance=tol)
                add(out, cases, "%s_a%s_%s" % (kname, spin, tname), kind, spin, opt, camera_fan(seed, 22))
    # every step forced: a tolerance nothing meets (60 steps of 1e-5)
    st = np.array([[0, 6.0, 1.2, 0.3, -1, -0.8, 1.5, 2.5], [0, 2.2, 1.5, 0.0, -1, -0.2, 0.5, 3.0],
                   [0, 30.0, 0.8, 1.0, -1, -1.0, -2.0, 4.0], [0, 12.0, 2.0, 0.5, -1, -0.9, 0.3, -3.0]])
    add(out, cases, "ks_a0.9_forced", po.KERR_KS, 0.9, po.options(max_steps=60, tolerance=1e-30), st)
    add(out, cases, "bl_a0.9_forced", po.KERR_BL, 0.9, po.options(max_steps=60, tolerance=1e-30), st)
    # Boyer-Lindquist into the horizon: radial plunges and near-critical spirals at high spin
    h = np.pi / 2
    st = np.array([[0, 8.0, h, 0.0, -1, -1.0, 0.0, 0.0], [0, 8.0, h, 0.0, -1, -1.0, 0.0, 1.5],
                   [0, 8.0, 1.2, 0.0, -1, -1.0, 0.8, -1.0], [0, 5.0, h, 0.0, -1, -0.9, 0.0, 2.0],
                   [0, 6.0, 0.9, 1.0, -1, -1.0, 1.0, 0.5], [0, 4.0, h, 0.0, -1, -1.0, 0.0, -3.0]])
    add(out, cases, "bl_a0.999_plunge", po.KERR_BL, 0.999, po.options(max_steps=1500, tolerance=1e-8), st)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "rays_v2.npz"), **out)
    n = sum(out[c + "_in"].shape[0] for c in cases)
    terms = np.concatenate([out[c + "_term"] for c in cases])
    print("cases:", len(cases), "rays:", n, "classes:", {int(k): int((terms == k).sum()) for k in np.unique(terms)})


if __name__ == "__main__":
    main()
