"""Generates the committed golden vectors from the CPU oracle (oracle/).

The reference's own path cannot run here (Rust; no cargo/rustc/wasm-pack) and
the reference holds no golden trajectories, so these vectors are
the oracle's outputs on fixed inputs: they pin the oracle against accidental
change (CPU test) and are what the HIP path is compared with on the GPU box.

    python tests/golden/make_golden.py      # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402


def ray_set(seed, n):
    rng = np.random.default_rng(seed)
    st = np.zeros((n, 8))
    st[:, 1] = rng.uniform(4.0, 80.0, n)
    st[:, 2] = rng.uniform(0.15, np.pi - 0.15, n)
    st[:, 3] = rng.uniform(0.0, 2 * np.pi, n)
    st[:, 4] = -1.0
    st[:, 5] = rng.uniform(-1.0, 0.3, n)
    st[:, 6] = rng.uniform(-7.0, 7.0, n)
    st[:, 7] = rng.uniform(-7.0, 7.0, n)
    return st


def known_scenarios():
    """Initial states used by the reference's doc-test / legacy tests."""
    h = np.pi / 2
    return np.array([
        [0, 20.0, h, 0, -1.0, -1.0, 0.0, 3.5],     # geodesic/mod.rs:175 doc-test
        [0, 20.0, 1.57, 0, -1.0, -1.0, 0.0, 3.5],  # _legacy_src/integrator.rs:111
        [0, 3.0, 1.57, 0, -1.0, -1.0, 0.0, 0.0],   # _legacy_src/integrator.rs:361
        [0, 5.0, 1.57, 0, -1.0, -0.5, 0.0, 2.0],
        [0, 10.0, 0.3, 1.0, -1.0, -1.0, 1.0, 0.5],
        [0, 50.0, 2.5, 2.0, -1.0, -1.0, -2.0, -4.0],
    ], dtype=np.float64)


def main():
    out = {}
    cases = []
    for kind, kname in ((po.KERR_KS, "ks"), (po.KERR_BL, "bl"), (po.SCHWARZSCHILD, "schw")):
        for spin in (0.0, 0.5, 0.9, 0.998):
            if kind == po.SCHWARZSCHILD and spin != 0.0:
                continue
            for method, mname, opt in (
                    (po.METHOD_RKF45, "rkf45", po.options(max_steps=2048, tolerance=1e-8)),
                    (po.METHOD_RKF45, "rkf45t9", po.options(max_steps=2048, tolerance=1e-9)),
                    (po.METHOD_SYMPLECTIC, "symp",
                     po.options(method=po.METHOD_SYMPLECTIC, step_size=0.05, max_steps=1024)),
                    (po.METHOD_RK4, "rk4", po.options(method=po.METHOD_RK4, step_size=0.05, max_steps=1024))):
                if mname == "rkf45t9" and not (kind == po.KERR_KS and spin == 0.998):
                    continue
                n = 48 if method == po.METHOD_RKF45 else 24
                st = np.concatenate([known_scenarios(), ray_set(1000 + len(cases), n)])
                m = po.metric(kind, 1.0, spin)
                res = po.integrate_batch(m, opt, st, nthreads=4)
                key = f"{kname}_a{spin}_{mname}"
                cases.append(key)
                out[key + "_in"] = st
                out[key + "_out"] = res["states"]
                out[key + "_steps"] = res["steps"]
                out[key + "_term"] = res["term"]
                out[key + "_drift"] = res["drift"]
                out[key + "_meta"] = np.array([kind, spin, method, opt.tolerance, opt.max_steps,
                                               opt.step_size, opt.escape_radius,
                                               opt.renormalize_interval, opt.initial_step])
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "rays_v1.npz"), **out)

    # small frame fixture (config C3 semantics at 64x36): camera r0 = 60 M, theta = 97 deg
    W, H = 64, 36
    th = np.deg2rad(97.0)
    eye = (60.0 * np.sin(th), 60.0 * np.cos(th), 0.0)
    cam = po.camera_look_at(eye, aspect=W / H)
    fp = po.frame_params(W, H, spin=0.999)
    lut = po.blackbody_lut(fp.lut_width, fp.lut_height, fp.lut_max_temp)
    fr = po.render_frame(cam, fp, lut, nthreads=4)
    st = fr["stats"]
    np.savez_compressed(
        os.path.join(HERE, "frame_v1.npz"), eye=np.array(eye), width=W, height=H, spin=0.999,
        rgba=fr["rgba"], states=fr["states"], steps=fr["steps"], term=fr["term"],
        drift=fr["drift"], accepted_steps=st.accepted_steps, rkf_tries=st.rkf_tries,
        term_count=np.array(list(st.term_count)), crossings=st.crossings,
        lut_probe=lut.reshape(64, 512, 4)[::9, ::37].copy(),
        pixel_states=np.array([po.pixel_state(cam, W, H, i, j) for j in (0, 17, 35) for i in (0, 31, 63)]))
    print("cases:", len(cases), "frame steps:", st.accepted_steps)


if __name__ == "__main__":
    main()
