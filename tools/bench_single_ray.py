"""Latency of the reference's path entry, integrate_ray_relativistic (lib.rs:422-464), one ray per
call: a serial chain on one GPU lane.  Reports the per-call wall time for the doc-test ray
(~220 accepted steps), for 0-step and 1-step calls (fixed overhead of the fused launch: one kernel
launch + the PCIe write of the result the host polls) and the per-ray time of the same rays in
batches.  Run on the GPU box: python tools/bench_single_ray.py"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import blackhole_simulation_amd as bh  # noqa: E402

if __name__ == "__main__":
    v = np.array([0, 20.0, np.pi / 2, 0, -1.0, -1.0, 0.0, 3.5])
    with bh.PhysicsEngine(1.0, 0.9) as e:
        res = {}
        for steps, label in ((10000, "doc-test ray to termination"), (1, "one step"), (0, "zero steps (call overhead)")):
            e.integrate_ray_relativistic(v, steps, 1e-8, True)
            t = time.perf_counter()
            n = 500
            for _ in range(n):
                e.integrate_ray_relativistic(v, steps, 1e-8, True)
            dt = (time.perf_counter() - t) / n
            res[steps] = dt
            clk = (C.c_uint64 * 3)()
            e._lib.grv_last_ray_clocks(e._h, C.byref(clk))
            if clk[2]:
                print(json.dumps({"call": "integrate_ray_relativistic", "case": label + ": the try loop on the device clocks",
                                  "tries": int(clk[2]), "shader_cycles": int(clk[0]), "us_100MHz_counter": clk[1] / 100.0,
                                  "shader_MHz": round(clk[0] / max(clk[1] / 100.0, 1e-9), 1),
                                  "cycles_per_try": round(clk[0] / clk[2], 1)}), flush=True)
            print(json.dumps({"call": "integrate_ray_relativistic", "case": label, "us_per_call": round(dt * 1e6, 1)}), flush=True)
        print(json.dumps({"call": "integrate_ray_relativistic", "case": "per accepted step (224 steps, overhead removed)",
                          "us_per_step": round((res[10000] - res[0]) / 224 * 1e6, 3)}), flush=True)
        # the same entry under the FAST contract (grv_engine_set_ray_arith): a third of the instructions
        e.set_ray_arith(bh.ARITH_FAST)
        e.integrate_ray_relativistic(v, 10000, 1e-8, True)
        t = time.perf_counter()
        for _ in range(500):
            e.integrate_ray_relativistic(v, 10000, 1e-8, True)
        dtf = (time.perf_counter() - t) / 500
        clk = (C.c_uint64 * 3)()
        e._lib.grv_last_ray_clocks(e._h, C.byref(clk))
        print(json.dumps({"call": "integrate_ray_relativistic", "case": "doc-test ray to termination, FAST contract",
                          "us_per_call": round(dtf * 1e6, 1), "tries": int(clk[2]),
                          "cycles_per_try": round(clk[0] / max(clk[2], 1), 1),
                          "speedup_over_strict": round(res[10000] / dtf, 2)}), flush=True)
        e.set_ray_arith(bh.ARITH_STRICT)
        for nb in (1, 64, 4096, 262144):
            st = np.tile(v, (nb, 1))
            st[:, 7] = np.linspace(3.0, 4.0, nb) if nb > 1 else 3.5
            for arith, an in ((bh.ARITH_STRICT, "strict"), (bh.ARITH_FAST, "fast")):
                o = bh.engine.default_options(max_steps=10000, arith=arith)
                e.integrate_batch(st, o)
                t = time.perf_counter()
                reps = 20 if nb < 100000 else 3
                for _ in range(reps):
                    r = e.integrate_batch(st, o)
                dt = (time.perf_counter() - t) / reps
                print(json.dumps({"call": "integrate_batch (host buffers)", "rays": nb, "arith": an,
                                  "ms_per_call": round(dt * 1e3, 3), "us_per_ray": round(dt / nb * 1e6, 3),
                                  "mean_steps": float(r["steps"].mean())}), flush=True)
