"""Parity at BASELINE's full size: every ray of the 3840x2160 bench frame (a = 0.999, RKF45 tol
1e-8, <= 2048 steps) integrated by the HIP engine (STRICT and FAST contracts) and by the CPU
oracle, compared ray by ray (~30 s of host time on 16 cores): STRICT must match bit for bit, FAST to rounding.  With GRV_PARITY_JSON=<path> the
comparison is also written out (profiles/r01_full_frame_parity.json was made that way)."""
import ctypes as C
import json
import os
import time

import numpy as np
import pytest

W, H = (int(v) for v in os.environ.get("GRV_PARITY_SIZE", "3840x2160").split("x"))   # 7680x4320: the maximum size BASELINE names
TOL = float(os.environ.get("GRV_PARITY_TOL", "1e-8"))   # 1e-9: BASELINE configs[4], the numerical-parity run
EYE = (60.0 * np.sin(np.deg2rad(97.0)), 60.0 * np.cos(np.deg2rad(97.0)), 0.0)


def _threads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, n)


@pytest.mark.gpu
def test_every_ray_of_the_bench_frame(engine_mod, oracle):
    import torch
    bh, po = engine_mod, oracle
    n = W * H
    t = time.time()
    ref = po.render_frame(po.camera_look_at(EYE, aspect=W / H),
                          po.frame_params(W, H, spin=0.999, opt=po.options(max_steps=2048, tolerance=TOL)), None,
                          nthreads=_threads())
    out = {"frame": "%dx%d a=0.999 RKF45 tol=%g max_steps=2048" % (W, H, TOL), "rays": n,
           "oracle_seconds": round(time.time() - t, 1), "oracle_threads": _threads(),
           "oracle_accepted_steps": int(ref["steps"].sum())}
    try:
        peak = float(ref["rgba"][..., :3].max())
        with bh.PhysicsEngine(1.0, 0.999) as e:
            cam = bh.camera_look_at(EYE, aspect=W / H)
            for name, arith in (("strict", bh.ARITH_STRICT), ("fast", bh.ARITH_FAST)):
                p = bh.render_params(W, H, arith=arith, tolerance=TOL)
                rgba = torch.zeros(n, 4, dtype=torch.float32, device="cuda:0")
                fs = torch.zeros(n, 8, dtype=torch.float64, device="cuda:0")
                steps = torch.zeros(n, dtype=torch.int32, device="cuda:0")
                term = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
                drift = torch.zeros(n, dtype=torch.float64, device="cuda:0")
                e.render_frame_device(cam, p, rgba, fs, steps, term, drift)
                torch.cuda.synchronize()
                a, b = fs.cpu().numpy(), ref["states"]
                err = (np.abs(a - b) / np.maximum(1.0, np.abs(b))).max(axis=1)
                st = steps.cpu().numpy().astype(np.int64)
                ds = np.abs(st - ref["steps"].astype(np.int64))
                cls = term.cpu().numpy() != ref["term"]
                dpx = float(np.abs(rgba.cpu().numpy() - ref["rgba"].reshape(-1, 4)).max())
                out[name] = {
                    "accepted_steps": int(st.sum()),
                    "termination_class_mismatches": int(cls.sum()),
                    "step_count_mismatches": int((ds > 0).sum()), "max_step_count_difference": int(ds.max()),
                    "endpoint_rel_err": {"p50": float(np.median(err)), "p99": float(np.percentile(err, 99)),
                                         "p99.99": float(np.percentile(err, 99.99)), "max": float(err.max())},
                    "rays_above_1e-6": int((err > 1e-6).sum()),
                    "pixel_max_abs_diff_over_peak": dpx / peak,
                    "rays_with_any_bit_different": int(((a != b) & ~(np.isnan(a) & np.isnan(b))).any(axis=1).sum()),
                    "drift_values_different": int((drift.cpu().numpy() != ref["drift"]).sum()),
                    "pixels_with_any_bit_different": int((rgba.cpu().numpy() != ref["rgba"].reshape(-1, 4)).any(axis=1).sum()),
                }
                same = ds == 0
                big = np.flatnonzero(err > 1e-5)
                out[name]["rays_above_1e-5"] = [
                    {"ray": int(i), "pixel": [int(i % W), int(i // W)], "rel_err": float(err[i]), "steps_oracle": int(ref["steps"][i]),
                     "step_count_difference": int(st[i] - int(ref["steps"][i])), "class_mismatch": bool(cls[i])}
                    for i in big[:64]]
                if arith == bh.ARITH_STRICT:
                    # reference operation order + the specified sin/cos/pow: the same bits as the
                    # checker for every end state, step count, class, drift and pixel of the frame
                    assert cls.sum() == 0 and ds.max() == 0
                    assert out[name]["rays_with_any_bit_different"] == 0
                    assert out[name]["drift_values_different"] == 0
                    assert out[name]["pixels_with_any_bit_different"] == 0
                else:   # FAST: rounding may flip an accept / reject decision of the controller on a few
                        # rays (one more step, or the same count through a different h history)
                    assert cls.sum() <= 2 and (~same).sum() <= 1e-5 * n and big.size <= 1e-6 * n
                    # ... and the few rays beyond 1e-5 are NAMED, BOUNDED and EXPLAINED, not merely "few": each
                    # is a ray whose accept / reject history departed from the oracle's (one more step, or the
                    # same count through another h sequence), so its LAST step lands elsewhere on the SAME
                    # geodesic (|h| <= 10 out at r ~ 1000: up to 1e-2 relative in t and r).  Bounded: <= 5e-2;
                    # explained: with the displacement along the ray taken out -- d lambda from the t
                    # components, tangent = the oracle's get_state_derivative at its end state -- what is
                    # left is the difference of two RKF45 step sequences along one geodesic (each within its
                    # tol = 1e-8 per step): 9.3e-6 and 2.0e-8 measured for the two rays, held to 5e-5.
                    assert float(err[big].max(initial=0.0)) <= 5e-2
                    m_ks = po.metric(po.KERR_KS, 1.0, 0.999)
                    for k, i in enumerate(big):
                        sb = po.make_state(list(b[i]))
                        dv = po.lib().orc_state_derivative(C.byref(sb), C.byref(m_ks))
                        tangent = np.array(list(dv.x) + list(dv.p))
                        dlam = (a[i, 0] - b[i, 0]) / tangent[0]
                        resid = np.abs(a[i] - (b[i] + dlam * tangent)) / np.maximum(1.0, np.abs(b[i]))
                        if k < 64:
                            out[name]["rays_above_1e-5"][k].update(d_lambda=float(dlam), rel_err_off_the_ray=float(resid.max()))
                        assert abs(dlam) <= 10.0 * (1.0 + 1e-6) and resid.max() <= 5e-5, (int(i), float(dlam), resid)
                # (FAST's median sits at 1.9e-10 at tol 1e-8 and 1.3e-9 at tol 1e-9 -- a third more steps per
                # ray and smaller ones, so the rounding of the step-size controller weighs more; STRICT: 0)
                assert np.median(err) <= (1e-9 if TOL >= 1e-8 else 1e-8) and dpx <= 1e-5 * peak
    finally:  # the record is written even when a bar above fails: the failing rays are in it
        path = os.environ.get("GRV_PARITY_JSON")
        if path:
            with open(path, "w") as f:
                json.dump(out, f, indent=1)
