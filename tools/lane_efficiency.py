"""How many lane-tries the headline frame wastes on intra-wave divergence: a wave (one 8x8 pixel
block) runs until its slowest ray is done, so lane efficiency = sum(steps) / (64 * max(steps))
per block, step-weighted over the frame.  Run on the GPU box: python tools/lane_efficiency.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import blackhole_simulation_amd as bh  # noqa: E402

if __name__ == "__main__":
    W, H = (3840, 2160) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
    eye = (60.0 * np.sin(np.deg2rad(97.0)), 60.0 * np.cos(np.deg2rad(97.0)), 0.0)
    with bh.PhysicsEngine(1.0, 0.999) as e:
        cam = bh.camera_look_at(eye, aspect=W / H)
        p = bh.render_params(W, H, max_steps=2048, shading=0, arith=bh.ARITH_FAST)
        steps = torch.zeros(H, W, dtype=torch.int32, device="cuda:0")
        e.render_frame_device(cam, p, steps=steps)
        torch.cuda.synchronize()
        st = e.frame_stats()
    s = steps.to(torch.float64)
    out = {"width": W, "height": H, "accepted_steps": int(s.sum().item())}
    for bw, bh_ in ((8, 8), (64, 1), (16, 4), (4, 16)):
        hh, ww = H // bh_ * bh_, W // bw * bw
        b = s[:hh, :ww].reshape(hh // bh_, bh_, ww // bw, bw).permute(0, 2, 1, 3).reshape(-1, bw * bh_)
        mx = b.max(dim=1).values
        out["lane_eff_%dx%d" % (bw, bh_)] = round(float(b.sum() / (mx.sum() * bw * bh_)), 4)
    # workgroup = 4 waves = 4 consecutive 8x8 blocks of a tile row: the block's slowest wave
    b = s[:H // 8 * 8, :W // 8 * 8].reshape(H // 8, 8, W // 8, 8).permute(0, 2, 1, 3).reshape(H // 8, W // 8, 64)
    wave_max = b.max(dim=2).values
    out["wave_max_mean"] = round(float(wave_max.mean()), 2)
    out["wave_max_p50_p99_max"] = [float(wave_max.quantile(q)) for q in (0.5, 0.99, 1.0)]
    if st is not None:
        out["rkf_tries"] = int(st.rkf_tries)
    print(json.dumps(out))
