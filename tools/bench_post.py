"""Secondary measurement: the post chain at 3840x2160 (TAA resolve, ATAA resolve, bloom).
Algorithmic HBM bytes per pixel (RGBA f32 = 16 B/px): TAA / ATAA read current + history and
write one frame = 48 B/px; bloom reads the scene twice (bright pass, combine) and writes one
frame, intermediates are 1/4 and 1/16 size: ~(16 + 16 + 16) + 16*(1/4)*... ~ 56 B/px.
Run on the GPU box: python tools/bench_post.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import blackhole_simulation_amd as bh  # noqa: E402

W, H = 3840, 2160


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


if __name__ == "__main__":
    cur = torch.rand(H, W, 4, device="cuda:0") * 4.0
    hist = torch.rand(H, W, 4, device="cuda:0") * 4.0
    out = torch.zeros_like(cur)
    px = W * H
    with bh.PhysicsEngine(1.0, 0.9) as e:
        cam = bh.camera_look_at((59.55, -7.31, 0.0), aspect=W / H)
        ap = bh.AtaaParams()
        ap.width, ap.height, ap.half_storage = W, H, 1
        iv = np.array(cam.inv_view).reshape(4, 4).T
        ip = np.array(cam.inv_proj).reshape(4, 4).T
        pvp = (np.linalg.inv(ip) @ np.linalg.inv(iv)).T.reshape(-1)
        for k in range(16):
            ap.inv_view[k], ap.inv_proj[k], ap.prev_view_proj[k] = cam.inv_view[k], cam.inv_proj[k], float(pvp[k])
        for k in range(3):
            ap.position[k] = cam.position[k]
        s = torch.cuda.current_stream().cuda_stream
        apf = bh.AtaaParams.from_buffer_copy(ap)
        apf.arith = bh.ARITH_FAST
        cases = [
            ("taa_resolve", 48, lambda: e.post_taa_resolve(W, H, cur, hist, out, stream=s)),
            ("ataa_resolve", 48, lambda: e.post_ataa_resolve(ap, cur, hist, out, stream=s)),
            ("bloom (bright + 2x(H,V) + combine)", 56, lambda: e.post_bloom(W, H, cur, out, stream=s)),
            ("taa_resolve FAST", 48, lambda: e.post_taa_resolve(W, H, cur, hist, out, stream=s, arith=bh.ARITH_FAST)),
            ("ataa_resolve FAST", 48, lambda: e.post_ataa_resolve(apf, cur, hist, out, stream=s)),
            ("bloom FAST", 56, lambda: e.post_bloom(W, H, cur, out, stream=s, arith=bh.ARITH_FAST)),
        ]
        for name, bpp, fn in cases:
            ms = timed(fn)
            print(json.dumps({"kernel": name, "width": W, "height": H, "ms": round(ms, 4),
                              "algorithmic_bytes_per_px": bpp,
                              "algorithmic_GBps": round(px * bpp / ms / 1e6, 1),
                              "frac_of_8TBps": round(px * bpp / ms / 1e6 / 8000.0, 3)}), flush=True)
