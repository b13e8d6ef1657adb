"""Lane efficiency of the f32 compute march on the config-4 frame (7680x4320, 1024-step budget): a
wave runs until its slowest ray is done, so efficiency = sum(steps) / (lanes * max(steps)) per wave,
step-weighted over the frame -- for the one-ray form (one 8x8 pixel block per wave) and the packed
form (two rays per lane: a 16x8 block per wave).  Also: how the steps are distributed (the shadow's
interior marches the whole budget, the sky ~100 steps).  Run on the GPU box."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import blackhole_simulation_amd as bh  # noqa: E402

if __name__ == "__main__":
    W, H = (7680, 4320) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
    ms = 1024 if len(sys.argv) < 4 else int(sys.argv[3])
    eye = (60.0 * np.sin(np.deg2rad(97.0)), 60.0 * np.cos(np.deg2rad(97.0)), 0.0)
    with bh.PhysicsEngine(1.0, 0.999) as e:
        cam = bh.camera_look_at(eye, aspect=W / H)
        wp = bh.wgsl_params(W, H, cam, 1.0, 0.999, max_steps=ms, arith=bh.ARITH_FAST_PACKED)
        rgba = torch.zeros(H * W, 4, dtype=torch.float32, device="cuda:0")
        steps = torch.zeros(H, W, dtype=torch.int32, device="cuda:0")
        e.render_frame_wgsl(wp, rgba, steps=steps)
        torch.cuda.synchronize()
    s = steps.to(torch.float64)
    out = {"width": W, "height": H, "max_steps": ms, "steps": int(s.sum().item()), "mean_steps_per_ray": round(float(s.mean()), 2)}
    q = [0.5, 0.9, 0.99, 0.999, 1.0]
    out["steps_per_ray_percentiles"] = dict(zip(map(str, q), [float(s.flatten().quantile(torch.tensor(x, dtype=torch.float64, device=s.device))) if s.numel() < 16e6 else float(np.quantile(s.flatten().cpu().numpy(), x)) for x in q]))
    out["rays_at_budget_frac"] = round(float((s >= ms).double().mean()), 5)
    for name, bw, bh_ in (("one_ray_8x8", 8, 8), ("packed_16x8", 16, 8)):
        hh, ww = H // bh_ * bh_, W // bw * bw
        b = s[:hh, :ww].reshape(hh // bh_, bh_, ww // bw, bw).permute(0, 2, 1, 3).reshape(-1, bw * bh_)
        mx = b.max(dim=1).values
        out["lane_eff_" + name] = round(float(b.sum() / (mx.sum() * bw * bh_)), 4)
        out["wave_steps_sum_" + name] = int(mx.sum().item())
    print(json.dumps(out))
