"""What one rank of an N-GPU strong split does per frame, measured on ONE GPU: the rank's tile
share (64x64 tiles k with k % N == r) of the bench frame, queued back to back with no host wait
exactly as bench.py's loop does (minus the RCCL gather, which overlaps the next frame).  From the
slowest rank's time per N: the strong-scaling efficiency the compute side allows,
t(1) / (N * max_r t_r(N)), and the host time of a frame call (the budget that must stay below the
device time for the queue never to run dry).
    python tools/bench_rank_share.py [c3|c4] [fast|packed] > profiles/r02_rank_share_<cfg>.jsonl"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import blackhole_simulation_amd as bh  # noqa: E402
from blackhole_simulation_amd import distributed as D  # noqa: E402

_R0, _TH = [float(x) for x in os.environ.get("RS_EYE", "60,97").split(",")]   # RS_EYE=R0,THETA: another camera of the sweep
EYE = (_R0 * np.sin(np.deg2rad(_TH)), _R0 * np.cos(np.deg2rad(_TH)), 0.0)


WGSL_ARITH = bh.ARITH_FAST


def share(e, cfg, W, H, world, rank, frames, two_streams=False):
    cam = bh.camera_look_at(EYE, aspect=W / H)
    params = bh.render_params(W, H, arith=bh.ARITH_FAST)
    rp = D.rank_params(params, world, rank)
    n = e.frame_ray_count(rp)
    wp = bh.wgsl_params(W, H, cam, 1.0, 0.999, max_steps=1024, arith=WGSL_ARITH,
                        tile_world=world, tile_rank=rank) if cfg == "c4" else None
    rgbas = [torch.zeros(n, 4, dtype=torch.float32, device="cuda:0") for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()] if two_streams else [torch.cuda.current_stream()] * 2
    stream = streams[0].cuda_stream

    def go(i=0):
        s = streams[i % 2].cuda_stream
        if cfg == "c3":
            e.render_frame_device(cam, rp, rgba=rgbas[i % 2], stream=s)
        else:
            e.render_frame_wgsl(wp, rgbas[i % 2], stream=s, want_total=False)

    go(0)
    go(1)
    torch.cuda.synchronize()
    e.frame_stats_reset(stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for i in range(frames):
        h0 = time.perf_counter()
        go(i)
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / frames
    st = e.frame_stats(stream)
    return dt * 1e3, host / frames * 1e6, st.accepted_steps / frames, n


if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
    W, H = (3840, 2160) if cfg == "c3" else (7680, 4320)
    frames = 10 if cfg == "c3" else 4
    if len(sys.argv) > 2 and sys.argv[2] == "packed":
        WGSL_ARITH = bh.ARITH_FAST_PACKED
    with bh.PhysicsEngine(1.0, 0.999) as e:
        e.stats_accumulate(True)
        base = None
        for world, two in ((1, False), (2, False), (4, False), (8, False), (1, True), (2, True), (4, True), (8, True)):
            rows = [share(e, cfg, W, H, world, r, frames, two) for r in range(world)]
            ms = [r[0] for r in rows]
            if world == 1 and not two:
                base = ms[0]
            print(json.dumps({
                "config": cfg, "arith": "packed" if WGSL_ARITH == bh.ARITH_FAST_PACKED else "fast", "frame": [W, H], "n_gpus": world, "frames_in_flight": 2 if two else 1,
                "ms_per_frame_by_rank": [round(x, 3) for x in ms],
                "slowest_rank_ms": round(max(ms), 3),
                "steps_by_rank": [int(r[2]) for r in rows],
                "host_us_per_frame_call": round(float(np.mean([r[1] for r in rows])), 1),
                "slots_per_rank": rows[0][3],
                "compute_side_efficiency": round(base / (world * max(ms)), 4),
                "Mray_steps_per_s_if_ranks_ran_concurrently": round(sum(r[2] for r in rows) / max(ms) / 1e3, 1),
            }), flush=True)
