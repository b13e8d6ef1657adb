"""Secondary measurement: what ray compaction buys on an incoherent batch.
The 1080p frame's rays are integrated as a batch (grv_integrate_batch_device) in image order
(neighbouring lanes take near-identical step counts) and in a random permutation (every wave
mixes 35..435-step rays), under the refill kernel (the default), relaunch + compaction with
segment lengths K = 16, 64, 256, and one launch without either.
Run on the GPU box: python tools/bench_batch.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import blackhole_simulation_amd as bh  # noqa: E402

W, H = 1920, 1080
EYE = (60.0 * np.sin(np.deg2rad(97.0)), 60.0 * np.cos(np.deg2rad(97.0)), 0.0)

if __name__ == "__main__":
    n = W * H
    with bh.PhysicsEngine(1.0, 0.999) as e:
        # initial states = the frame's rays after zero steps
        cam = bh.camera_look_at(EYE, aspect=W / H)
        p0 = bh.render_params(W, H, max_steps=0, shading=0)
        fs = torch.zeros(n, 8, dtype=torch.float64, device="cuda:0")
        e.render_frame_device(cam, p0, final_state=fs)
        torch.cuda.synchronize()
        perm = torch.randperm(n, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(1))
        out = torch.zeros_like(fs)
        steps = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        ref = None
        for order, states in (("image order", fs), ("random permutation", fs[perm].contiguous())):
            for K in (0, -4, -32, 16, 64, 256, 1 << 20):
                o = bh.engine.default_options(max_steps=2048, arith=bh.ARITH_FAST, segment_tries=K)
                e.integrate_batch_device(n, states, o, out, steps)
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(5):
                    e.integrate_batch_device(n, states, o, out, steps)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t) / 5 * 1e3
                tot = int(steps.sum().item())
                res = out if order == "image order" else out[torch.argsort(perm)]
                if ref is None:
                    ref = res.clone()
                same = bool(torch.equal(res, ref))  # results must not depend on K or on the order
                print(json.dumps({"order": order, "schedule": ("refill every %d tries" % (-K or 8)) if K <= 0 else
                                  ("compaction every %d tries" % K if K < (1 << 20) else "one launch"),
                                  "rays": n, "accepted_steps": tot, "ms": round(ms, 3),
                                  "Mray_steps_per_s": round(tot / ms / 1e3, 1),
                                  "bitwise_equal_to_first": same}), flush=True)
