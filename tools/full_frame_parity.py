"""Parity evidence at BASELINE's full size: every ray of the 3840x2160 bench frame (a = 0.999,
RKF45 tol 1e-8, <= 2048 steps) integrated by the HIP engine (STRICT and FAST contracts) and by
the CPU oracle, compared ray by ray.  ~1 minute of host time on 16 cores.
Run on the GPU box: python tools/full_frame_parity.py > gpurun_out/full_frame_parity.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402

import blackhole_simulation_amd as bh  # noqa: E402
import pyoracle as po  # noqa: E402

sys.path.insert(0, ROOT)
from bench import usable_cores  # noqa: E402

W, H = 3840, 2160
EYE = (60.0 * np.sin(np.deg2rad(97.0)), 60.0 * np.cos(np.deg2rad(97.0)), 0.0)

if __name__ == "__main__":
    n = W * H
    t = time.time()
    ref = po.render_frame(po.camera_look_at(EYE, aspect=W / H), po.frame_params(W, H, spin=0.999), None,
                          nthreads=usable_cores())
    cpu_s = time.time() - t
    out = {"frame": "%dx%d a=0.999 RKF45 tol=1e-8 max_steps=2048" % (W, H), "rays": n,
           "oracle_seconds": round(cpu_s, 1), "oracle_threads": usable_cores(),
           "oracle_accepted_steps": int(ref["steps"].sum())}
    with bh.PhysicsEngine(1.0, 0.999) as e:
        cam = bh.camera_look_at(EYE, aspect=W / H)
        for name, arith in (("strict", bh.ARITH_STRICT), ("fast", bh.ARITH_FAST)):
            p = bh.render_params(W, H, arith=arith)
            rgba = torch.zeros(n, 4, dtype=torch.float32, device="cuda:0")
            fs = torch.zeros(n, 8, dtype=torch.float64, device="cuda:0")
            steps = torch.zeros(n, dtype=torch.int32, device="cuda:0")
            term = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
            e.render_frame_device(cam, p, rgba, fs, steps, term)
            torch.cuda.synchronize()
            a, b = fs.cpu().numpy(), ref["states"]
            err = (np.abs(a - b) / np.maximum(1.0, np.abs(b))).max(axis=1)
            st = steps.cpu().numpy().astype(np.int64)
            ds = np.abs(st - ref["steps"].astype(np.int64))
            cls = term.cpu().numpy() != ref["term"]
            peak = float(ref["rgba"][..., :3].max())
            dpx = np.abs(rgba.cpu().numpy() - ref["rgba"].reshape(-1, 4)).max()
            out[name] = {
                "accepted_steps": int(st.sum()),
                "termination_class_mismatches": int(cls.sum()),
                "step_count_mismatches": int((ds > 0).sum()), "max_step_count_difference": int(ds.max()),
                "endpoint_rel_err": {"p50": float(np.median(err)), "p99": float(np.percentile(err, 99)),
                                     "p99.99": float(np.percentile(err, 99.99)), "max": float(err.max())},
                "rays_above_1e-6": int((err > 1e-6).sum()),
                "pixel_max_abs_diff_over_peak": float(dpx) / peak,
            }
    print(json.dumps(out, indent=1))
