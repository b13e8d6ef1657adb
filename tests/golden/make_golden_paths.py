"""Golden vectors for Trajectory.path (geodesic/mod.rs:150-161, IntegrationOptions.record_path):
the oracle's orc_integrate_path on the first six rays (the reference's doc-test / legacy scenarios)
of five ray sets of rays_v1.npz -> paths_v1.npz.  Rows are padded to the longest path with NaN.

    python tests/golden/make_golden_paths.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402

CASES = ["ks_a0.9_rkf45", "bl_a0.9_rkf45", "schw_a0.0_rkf45", "ks_a0.998_rk4", "ks_a0.5_symp"]


def main():
    z = np.load(os.path.join(HERE, "rays_v1.npz"))
    out = {"cases": np.array(CASES)}
    for key in CASES:
        kind, spin, method, tol, max_steps, step, esc, renorm, h0 = z[key + "_meta"]
        m = po.metric(int(kind), 1.0, float(spin))
        o = po.options(method=int(method), tolerance=float(tol), initial_step=float(h0), max_steps=int(max_steps),
                       escape_radius=float(esc), renormalize_interval=int(renorm), step_size=float(step))
        paths, counts = [], []
        for i in range(6):
            t, p = po.integrate_path(z[key + "_in"][i], m, o, cap=int(max_steps) + 1)
            assert p.shape[0] == t.steps_taken + 1
            paths.append(p)
            counts.append(p.shape[0])
        L = max(counts)
        arr = np.full((6, L, 8), np.nan)
        for i, p in enumerate(paths):
            arr[i, :p.shape[0]] = p
        out[key + "_paths"] = arr
        out[key + "_counts"] = np.array(counts, np.uint32)
    np.savez_compressed(os.path.join(HERE, "paths_v1.npz"), **out)
    print({k: out[k + "_counts"].tolist() for k in CASES})


if __name__ == "__main__":
    main()
