#!/usr/bin/env python
"""Per-ray cost of a 4096-ray batch from JavaScript (napi/bulk.js -> integrate_batch of the N-API
addon) beside the same call through ctypes, on the GPU box.  One JSON line on stdout
(-> profiles/r03_napi_bulk.json)."""
import json
import math
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import blackhole_simulation_amd as bh  # noqa: E402


def rays(n):  # napi/bulk.js rays()
    s = np.zeros((n, 8))
    for i in range(n):
        u = (i + 0.5) / n
        s[i] = [0, 20 + 40 * ((i * 7919) % n) / n, math.pi / 2 - 0.4 + 0.8 * u, 0.1 * i, -1, -1, 0.3 - 0.6 * u, -8 + 16 * u]
    return s


def main():
    node = os.environ.get("NODE", "node")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "bulk.json")
        subprocess.run([node, os.path.join(ROOT, "napi", "bulk.js"), out], check=True, capture_output=True, timeout=600, cwd=ROOT)
        js = json.load(open(out))
    n = js["batch"]["n"]
    st = rays(n)
    with bh.PhysicsEngine(1.0, 0.9) as e:
        o = bh.engine.default_options(max_steps=2000, tolerance=1e-8)
        e.integrate_batch(st, o)
        t0 = time.perf_counter()
        for _ in range(5):
            r = e.integrate_batch(st, o)
        ct_ms = (time.perf_counter() - t0) / 5 * 1e3
    same = bool(np.array_equal(np.array(js["batch"]["states"]).reshape(n, 8), r["states"]))
    print(json.dumps({"rays_per_call": n, "contract": "STRICT, RKF45 tol 1e-8, max 2000 steps, a = 0.9",
                      "js_integrate_batch_ms": round(js["batch_ms"], 3), "js_us_per_ray": round(js["us_per_ray_batch"], 4),
                      "ctypes_integrate_batch_ms": round(ct_ms, 3), "ctypes_us_per_ray": round(ct_ms * 1e3 / n, 4),
                      "js_over_ctypes": round(js["batch_ms"] / ct_ms, 3),
                      "js_one_ray_entry_us_per_ray": round(js["single_ms_per_ray"] * 1e3, 1),
                      "same_bits_js_and_ctypes": same,
                      "accepted_steps": int(np.sum(r["steps"]))}))


if __name__ == "__main__":
    main()
