"""Runs on the GPU box: renders the c2 GLSL FAST frame (default preset, 1920x1080, 512 steps) with the
library named on the command line and saves colours + step counts; with two .npz names, compares them
bit for bit.  Used to show that a control-flow rewrite of the march leaves every pixel unchanged."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 3 and sys.argv[1].endswith(".npz") and sys.argv[2].endswith(".npz"):
    a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
    print("steps identical:", bool(np.array_equal(a["steps"], b["steps"])), " pixels identical:",
          bool(np.array_equal(a["rgba"].view(np.uint32), b["rgba"].view(np.uint32))),
          " differing pixels:", int((a["rgba"].view(np.uint32) != b["rgba"].view(np.uint32)).any(-1).sum()))
    sys.exit(0)
import torch  # noqa: E402

import blackhole_simulation_amd as bh  # noqa: E402
W, H = 1920, 1080
out = {}
with bh.PhysicsEngine(1.0, 0.999) as e:
    for name, kw in (("default", {}), ("time", {"time": 12.5}), ("march_disk", {"features": 7, "turbulence": 0.75})):
        gp = bh.glsl_params(W, H, 1.0, 0.999, max_ray_steps=512, arith=1, **kw)
        rgba = torch.zeros(W * H, 4, dtype=torch.float32, device="cuda:0")
        steps = torch.zeros(W * H, dtype=torch.int32, device="cuda:0")
        e.render_frame_glsl(gp, rgba, steps)
        out[name] = (rgba.cpu().numpy(), steps.cpu().numpy())
np.savez(sys.argv[1], rgba=np.stack([v[0] for v in out.values()]), steps=np.stack([v[1] for v in out.values()]))
