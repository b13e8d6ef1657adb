"""The oracle must reproduce the committed golden vectors bit-for-bit: guards the checker
itself against silent change.  The ray path's sin/cos/pow are the written-out routines of
oracle/ref_libm.c -- the camera's acos / atan2 / sin / cos and the look-at helper's tan included --
so none of the vectors depends on the host's libm (glibc's sincos() and cos() differ in the last
bit for some arguments, and a compiler may or may not merge a sin / cos pair into sincos())."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases():
    z = np.load(os.path.join(GOLD, "rays_v1.npz"))
    return z, [str(c) for c in z["cases"]]


def test_golden_files_present():
    assert os.path.exists(os.path.join(GOLD, "rays_v1.npz"))
    assert os.path.exists(os.path.join(GOLD, "frame_v1.npz"))
    assert os.path.exists(os.path.join(GOLD, "make_golden.py"))
    assert os.path.exists(os.path.join(GOLD, "paths_v1.npz")) and os.path.exists(os.path.join(GOLD, "make_golden_paths.py"))


def test_oracle_reproduces_ray_golden(oracle):
    z, cases = _cases()
    assert len(cases) >= 20
    for key in cases:
        kind, spin, method, tol, max_steps, step, esc, renorm, h0 = z[key + "_meta"]
        m = oracle.metric(int(kind), 1.0, float(spin))
        opt = oracle.options(method=int(method), tolerance=float(tol), initial_step=float(h0),
                             max_steps=int(max_steps), escape_radius=float(esc),
                             renormalize_interval=int(renorm), step_size=float(step))
        res = oracle.integrate_batch(m, opt, z[key + "_in"], nthreads=2)
        assert np.array_equal(res["term"], z[key + "_term"]), key
        assert np.array_equal(res["steps"], z[key + "_steps"]), key
        assert np.array_equal(res["states"], z[key + "_out"], equal_nan=True), key
        assert np.array_equal(res["drift"], z[key + "_drift"], equal_nan=True), key


def test_oracle_reproduces_frame_golden(oracle):
    z = np.load(os.path.join(GOLD, "frame_v1.npz"))
    W, H = int(z["width"]), int(z["height"])
    cam = oracle.camera_look_at(tuple(z["eye"]), aspect=W / H)
    fp = oracle.frame_params(W, H, spin=float(z["spin"]))
    fr = oracle.render_frame(cam, fp, None, nthreads=4)
    assert fr["stats"].accepted_steps == int(z["accepted_steps"])
    assert np.array_equal(fr["term"], z["term"])
    assert np.array_equal(fr["steps"], z["steps"])
    assert np.array_equal(fr["rgba"], z["rgba"])
    assert np.array_equal(fr["states"], z["states"], equal_nan=True)
    ps = np.array([oracle.pixel_state(cam, W, H, i, j) for j in (0, 17, 35) for i in (0, 31, 63)])
    assert np.array_equal(ps, z["pixel_states"])


def test_single_ray_ffi_matches_batch(oracle):
    z, _ = _cases()
    key = "ks_a0.9_rkf45"
    v = z[key + "_in"][0]
    out = oracle.integrate_ray_relativistic(1.0, 0.9, v, 2048, 1e-8, True)
    # the FFI fixes h0=0.01, escape=1000, renorm=10 (gravitas-wasm/src/lib.rs:444-452)
    np.testing.assert_array_equal(out, z[key + "_out"][0])


def test_oracle_reproduces_path_golden(oracle):
    """Trajectory.path fixtures (tests/golden/make_golden_paths.py): the oracle still returns every
    recorded point bit for bit; point 0 is the input state (mod.rs:193-197), the last the end state."""
    z = np.load(os.path.join(GOLD, "rays_v1.npz"))
    pz = np.load(os.path.join(GOLD, "paths_v1.npz"))
    for key in [str(c) for c in pz["cases"]]:
        kind, spin, method, tol, max_steps, step, esc, renorm, h0 = z[key + "_meta"]
        m = oracle.metric(int(kind), 1.0, float(spin))
        o = oracle.options(method=int(method), tolerance=float(tol), initial_step=float(h0), max_steps=int(max_steps),
                           escape_radius=float(esc), renormalize_interval=int(renorm), step_size=float(step))
        for i in range(6):
            t, p = oracle.integrate_path(z[key + "_in"][i], m, o, cap=int(max_steps) + 1)
            n = int(pz[key + "_counts"][i])
            assert p.shape[0] == n == int(z[key + "_steps"][i]) + 1
            assert np.array_equal(p, pz[key + "_paths"][i, :n], equal_nan=True), (key, i)
            assert np.array_equal(p[0], z[key + "_in"][i])
            if n > 1:
                assert np.array_equal(p[-1], z[key + "_out"][i], equal_nan=True)
