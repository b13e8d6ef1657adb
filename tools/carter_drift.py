"""Oracle-independent accuracy of the two arithmetic contracts on the bench frame: the Carter
constant Q = p_theta^2 + cos^2(theta) (p_phi^2 / sin^2(theta) - a^2 p_t^2) of a null geodesic is
conserved exactly by the equations and only to truncation + rounding by an integrator.  For every
ray of the 3840x2160 frame: |Q_end - Q_start| / max(1, |Q_start|) under STRICT and under FAST (same
tolerance 1e-8), plus the |H| drift both kernels track.  Run on the GPU box: python tools/carter_drift.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import blackhole_simulation_amd as bh  # noqa: E402

W, H, A = 3840, 2160, 0.999


def carter(s):
    th, pt, pth, pph = s[:, 2], s[:, 4], s[:, 6], s[:, 7]
    c2, s2 = np.cos(th) ** 2, np.maximum(np.sin(th) ** 2, 1e-300)
    return pth ** 2 + c2 * (pph ** 2 / s2 - A * A * pt ** 2)


if __name__ == "__main__":
    eye = (60.0 * np.sin(np.deg2rad(97.0)), 60.0 * np.cos(np.deg2rad(97.0)), 0.0)
    n = W * H
    out = {"frame": "%dx%d a=%.3f RKF45 tol=1e-8 max_steps=2048" % (W, H, A), "rays": n}
    with bh.PhysicsEngine(1.0, A) as e:
        cam = bh.camera_look_at(eye, aspect=W / H)
        fs = torch.zeros(n, 8, dtype=torch.float64, device="cuda:0")
        drift = torch.zeros(n, dtype=torch.float64, device="cuda:0")
        e.render_frame_device(cam, bh.render_params(W, H, max_steps=0, shading=0), None, fs)
        torch.cuda.synchronize()
        q0 = carter(fs.cpu().numpy())
        for name, arith in (("strict", bh.ARITH_STRICT), ("fast", bh.ARITH_FAST)):
            e.render_frame_device(cam, bh.render_params(W, H, arith=arith, shading=0), None, fs, None, None, drift)
            torch.cuda.synchronize()
            dq = np.abs(carter(fs.cpu().numpy()) - q0) / np.maximum(1.0, np.abs(q0))
            dh = drift.cpu().numpy()
            out[name] = {"carter_rel_drift": {k: float(np.percentile(dq, p)) for k, p in (("p50", 50), ("p99", 99), ("p99.99", 99.99), ("max", 100))},
                         "max_abs_H": {k: float(np.percentile(dh, p)) for k, p in (("p50", 50), ("p99", 99), ("max", 100))}}
    print(json.dumps(out, indent=1))
