#!/usr/bin/env python
"""Headline benchmark: Mray-steps/s of the Kerr geodesic ray-marching path.

Workload (BASELINE.json configs[2], the one the metric is quoted on):
  3840x2160 per GPU, a = 0.999 Kerr-Schild, adaptive RKF45 tol 1e-8, <= 2048 steps,
  + Planck (T x g) LUT redshift shading; camera r0 = 60 M, theta = 97 deg, fov 60 deg.
A "step" is one frame: pixel->state init, integrate, shade (all on the GPU, outputs
resident in HBM), and for N > 1 the single gather of finished tiles to rank 0.

N > 1 (weak scaling): the image plane grows to (3840*gx) x (2160*gy), gx*gy = N, cut in
64x64 tiles dealt round-robin to the ranks, so every GPU integrates one 4K frame's
worth of rays.  --scaling strong splits the one 3840x2160 frame instead.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

GRID = {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
B_STEP, B_RAY = 144, 96  # algorithmic bytes: SURVEY.md 8(d) / DESIGN.md


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (a container on a 256-thread node is often limited to far fewer)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota|max> <period>"
            q, per = f.read().split()
            if q != "max":
                quota = int(q) / int(per)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, \
                    open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:  # cgroup v1
                q, per = int(f.read()), int(g.read())
                if q > 0:
                    quota = q / per
        except Exception:
            pass
    if quota:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def cpu_baseline(width, height, eye, target_seconds=15.0):
    """The oracle (C restatement of gravitas-core) timed on the host cores on a bounded,
    pixel-strided sample of the same workload.  Checker/baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    cores = usable_cores()
    cam = po.camera_look_at(eye, aspect=width / height)
    fp = po.frame_params(width, height, spin=0.999)
    lut = po.blackbody_lut(fp.lut_width, fp.lut_height, fp.lut_max_temp)
    t = time.time()
    probe = po.render_frame(cam, fp, lut, stride=(16, 16), nthreads=cores, want_states=False)
    dt = max(time.time() - t, 1e-3)
    rate = probe["stats"].accepted_steps / dt
    total = 183.0 * width * height  # ~steps in the full frame
    # the finest pixel stride whose estimated time stays inside the 10-30 s the sample is meant to take
    sx, sy = 32, 32
    for cand in ((1, 1), (2, 1), (2, 2), (3, 2), (3, 3), (4, 3), (4, 4), (6, 4), (6, 6), (8, 8), (12, 12),
                 (16, 16), (24, 24), (32, 32)):
        if total / (cand[0] * cand[1]) / rate <= 1.6 * target_seconds:
            sx, sy = cand
            break
    t = time.time()
    out = po.render_frame(cam, fp, lut, stride=(sx, sy), nthreads=cores, want_states=False)
    dt = time.time() - t
    st = out["stats"]
    # the same path on one core (BASELINE.md section 4 asks for both), on a 1/1024 subset
    t1 = time.time()
    one = po.render_frame(cam, fp, lut, stride=(32, 32), nthreads=1, want_states=False)
    one_rate = one["stats"].accepted_steps / max(time.time() - t1, 1e-3) / 1e6
    return {"value": round(st.accepted_steps / dt / 1e6, 4), "unit": "Mray-steps/s", "cores": cores,
            "one_core_value": round(one_rate, 4), "kind": "port",
            "sample": "C restatement of gravitas-core (Rust toolchain unavailable), OpenMP over "
                      "rays, 1/%d pixel-strided subset of the %dx%d frame: %d rays, %d accepted "
                      "steps in %.1f s" % (sx * sy, width, height, st.rays,
                                           st.accepted_steps, dt)}


def load_committed_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC
    passes (profiles/): measured off-line, never inside the timed region."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--arith", choices=["fast", "strict"], default="fast")
    ap.add_argument("--segment-tries", type=int, default=0)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: wait for each frame's gather before integrating the next frame")
    args = ap.parse_args()

    # stdout carries exactly one line, the JSON result: libraries that print banners on load
    # (RCCL prints its version / host / library path) get stderr until that line is written
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    import blackhole_simulation_amd as bh
    from blackhole_simulation_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    if not os.path.exists(bh.library_path()):
        if local_rank == 0:
            bh.build_library()  # checkout without the built artefact: compile it (hipcc)
        else:
            while not os.path.exists(bh.library_path()):
                time.sleep(1.0)
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "RANK" in os.environ  # under torchrun even a 1-rank job walks the RCCL path
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    gx, gy = GRID.get(world, (world, 1)) if args.scaling == "weak" else (1, 1)
    W, H = args.width * gx, args.height * gy
    th = np.deg2rad(97.0)
    eye = (60.0 * np.sin(th), 60.0 * np.cos(th), 0.0)
    arith = bh.ARITH_FAST if args.arith == "fast" else bh.ARITH_STRICT

    eng = bh.PhysicsEngine(1.0, 0.999, device=local_rank)
    cam = bh.camera_look_at(eye, aspect=W / H)
    params = bh.render_params(W, H, arith=arith, segment_tries=args.segment_tries, profile=1)
    rp = D.rank_params(params, world, rank)
    n_local = eng.frame_ray_count(rp)
    stream = torch.cuda.current_stream().cuda_stream
    # all buffers live outside the frame loop: the padded send buffer doubles as the render
    # target, rank 0 additionally holds the receive slots and the assembled image
    tg = D.TileGather(params, world, rank, 4, torch.float32, torch.device("cuda", local_rank)) \
        if use_dist else None
    overlap = tg is not None and not args.no_overlap
    if overlap:
        tg.enable_pipeline()  # second send buffer: frame i's gather runs under frame i+1's kernels
    buf = tg.local_view(n_local) if tg else torch.empty((n_local, 4), dtype=torch.float32, device="cuda")

    def dev_unpack(rparams, r, packed, image):
        eng.unpack_tiles_device(rparams, r, packed, image, 16, stream)

    def one_frame(i):
        target = tg.pipelined_view(i, n_local) if overlap else buf
        eng.render_frame_device(cam, rp, rgba=target, stream=stream)
        st = eng.frame_stats(stream)
        if overlap:
            tg.submit(i, dev_unpack, force_collective=True)  # finish frame i-1's exchange, start frame i's
        elif tg:
            tg.run(dev_unpack, force_collective=True)  # the one exchange: gather tiles -> rank 0
        return st

    def fence():
        if overlap:
            tg.drain(dev_unpack)  # the last frame's exchange is inside the timed region
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_frame(i)
    fence()
    t0 = time.perf_counter()
    steps_local = 0
    integ_ms = 0.0
    launches = 0
    max_drift = 0.0
    for i in range(args.steps):
        st = one_frame(args.warmup + i)
        max_drift = max(max_drift, st.max_drift)
        steps_local += st.accepted_steps
        integ_ms += st.integrate_ms
        launches += st.launches
    fence()
    elapsed = time.perf_counter() - t0

    agg = torch.tensor([elapsed, float(steps_local), float(n_local)], dtype=torch.float64, device="cuda")
    if use_dist:
        tmax = agg[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg[1:], op=dist.ReduceOp.SUM)
        elapsed = float(tmax.item())
    total_steps = float(agg[1].item())
    total_rays = float(agg[2].item())

    if rank == 0:
        value = total_steps / elapsed / 1e6
        # roofline of the dominant kernel (integrate_segment_kernel): algorithmic bytes of the
        # rays this rank integrated per launch / mean HIP-event duration of a launch
        per_frame_bytes = (steps_local / args.steps) * B_STEP + n_local * B_RAY
        launches_per_frame = max(launches / args.steps, 1.0)
        avg_launch_ms = integ_ms / max(launches, 1)
        achieved = (per_frame_bytes / launches_per_frame) / (avg_launch_ms * 1e-3) / 1e9
        traffic = load_committed_traffic()
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": (traffic or {}).get("hbm_bytes_per_launch"),
                    "kernel": "integrate_segment_kernel<KerrSchild,%s,RKF45>" % args.arith.upper(),
                    "avg_launch_ms": round(avg_launch_ms, 4),
                    "launches_per_frame": launches_per_frame,
                    "algorithmic_bytes_per_launch": int(per_frame_bytes / launches_per_frame)}
        if traffic and "valu" in traffic and args.arith == "fast" and not args.segment_tries:
            # the honest secondary picture (committed PMC pass): the register-resident kernel is
            # bound by FP64 VALU issue, not by HBM
            roofline["valu_issue_frac"] = traffic["valu"]["issue_frac"]
        line = {
            "metric": "Mray-steps/s", "value": round(value, 2), "unit": "Mray-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%dx%d frame (%dx%d per GPU x %d), a=0.999 Kerr-Schild, adaptive "
                                   "RKF45 tol=1e-8 h0=0.01 escape=1000 renorm=10 max_steps=2048, "
                                   "Planck LUT 512x64 Tmax=1e5 redshift shading, camera r0=60M "
                                   "theta=97deg fov=60deg" % (W, H, args.width, args.height, world),
                       "arith": args.arith, "segment_tries": args.segment_tries or "auto",
                       "partition": ("64x64 tiles round-robin, one gather to rank 0 per frame%s"
                                     % (", overlapped with the next frame" if overlap else ""))
                       if world > 1 else "single GPU",
                       "rays": int(total_rays), "accepted_steps_per_frame": int(total_steps / args.steps),
                       "max_hamiltonian_drift_rank0": max_drift},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.width, args.height, eye)
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)  # teardown chatter, if any, goes to stderr too

    eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
