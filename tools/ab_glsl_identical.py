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
if os.environ.get("AB_KERNEL") == "wgsl":  # the one-ray FAST compute march, stars on (the escape branch runs)
    import numpy as _np
    EYE = (60.0 * _np.sin(_np.deg2rad(97.0)), 60.0 * _np.cos(_np.deg2rad(97.0)), 0.0)
    with bh.PhysicsEngine(1.0, 0.999) as e:
        for name, spin, ms in (("a999", 0.999, 512), ("a5", 0.5, 150)):
            cam = bh.camera_look_at(EYE, aspect=W / H)
            gp = bh.wgsl_params(W, H, cam, 1.0, spin, max_steps=ms, arith=1, stars=1)
            e.update_params(1.0, spin)
            rgba = torch.zeros(W * H, 4, dtype=torch.float32, device="cuda:0")
            steps = torch.zeros(W * H, dtype=torch.int32, device="cuda:0")
            e.render_frame_wgsl(gp, rgba, steps)
            out[name] = (rgba.cpu().numpy(), steps.cpu().numpy())
    np.savez(sys.argv[1], rgba=np.stack([v[0] for v in out.values()]), steps=np.stack([v[1] for v in out.values()]))
    sys.exit(0)
with bh.PhysicsEngine(1.0, 0.999) as e:
    for name, kw in (("default", {}), ("time", {"time": 12.5}), ("march_disk", {"features": 7, "turbulence": 0.75}),
                     ("polar_60M", {"zoom": 60.0, "theta": 5.0}), ("polar_200M_time", {"zoom": 200.0, "theta": 5.0, "time": 3.25}),
                     ("close_10M", {"zoom": 10.0, "theta": 60.0})):   # views down the jets (tools/camera_sweep.py's cameras)
        theta = kw.pop("theta", None)
        gp = bh.glsl_params(W, H, 1.0, 0.999, max_ray_steps=512, arith=1, **kw)
        if theta is not None:
            gp.mouse[1] = theta / 180.0
        rgba = torch.zeros(W * H, 4, dtype=torch.float32, device="cuda:0")
        steps = torch.zeros(W * H, dtype=torch.int32, device="cuda:0")
        e.render_frame_glsl(gp, rgba, steps)
        out[name] = (rgba.cpu().numpy(), steps.cpu().numpy())
np.savez(sys.argv[1], rgba=np.stack([v[0] for v in out.values()]), steps=np.stack([v[1] for v in out.values()]))
