#!/usr/bin/env python
"""Headline benchmark: Mray-steps/s of the Kerr geodesic ray-marching path.

--config c3 (default; BASELINE.json configs[2], the one the metric is quoted on):
  3840x2160, a = 0.999 Kerr-Schild, adaptive RKF45 tol 1e-8, <= 2048 steps, f64,
  + Planck (T x g) LUT redshift shading; camera r0 = 60 M, theta = 97 deg, fov 60 deg.
--config c2 (BASELINE.json configs[1]): 1920x1080, a = 0.999, 512 max steps, one GPU, f32:
  --kernel glsl (default): the Cartesian Velocity-Verlet march of the WebGL fragment shader
  (fragment.glsl.ts:129-221 + chunks/metric.ts:96-149, the reference's production loop; the shader
  itself clamps the budget to 500), default "high-quality" preset; --kernel wgsl: the Kerr-Schild
  implicit-midpoint compute march (compute.wgsl.ts) at 512 steps.
--config c4 (BASELINE.json configs[3]):
  7680x4320, the f32 compute march (compute.wgsl.ts) at a fixed 1024-step budget: every
  ray marches until it terminates or has done 1024 steps.

--config c5 (BASELINE.json configs[4], the numerical-parity run):
  the c3 frame at RKF45 tol 1e-9 under the reference-order STRICT contract (the arithmetic whose
  end states, step counts and pixels equal the CPU oracle's bit for bit).

A "step" is one frame: pixel->state init, integrate, shade (all on the GPU, outputs resident
in HBM), and for N > 1 the single gather of finished tiles to rank 0 (RCCL over xGMI).

N > 1 (default --scaling strong, the split BASELINE's metric names: "at 3840x2160 ... 1/2/4/8
GPU"): the ONE frame is cut in 64x64 tiles dealt round-robin to the N ranks, so total work is
fixed and each rank integrates 1/N of the rays.  --scaling weak grows the image plane to
(W*gx) x (H*gy), gx*gy = N, instead (every GPU integrates one full frame's worth of rays).

Two hosts drive N > 1, both through the same C ABI and the same 64x64 round-robin tile deal:
  under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (WORLD_SIZE set), or a
      bare `python bench.py --gpus N` (which starts those N ranks itself): one process per GPU, one
      grv_engine each, the gather through torch.distributed's nccl backend (= RCCL) -- the default;
  `python bench.py --gpus N --launcher native`: ONE process, the C ABI's multi-GPU handle
      (grv_engine_create_multi: a host thread and two streams per device, one RCCL send/recv group
      per frame) -- the host BASELINE's north_star names; explicit until it has run on >= 2 devices.
Either line carries transport, rccl_version, rank_devices and the per-rank integrate times.

The frame loop holds no host wait: frames are queued back to back, the per-frame counters
accumulate on the device (grv_stats_accumulate) and are read once after the timed region.
At N = 1 the integrate launches are bracketed by HIP events recorded on the launch stream
(resolved after the loop); at N > 1 no events enter the timed loop and the roofline block
comes from a few profiled frames run after it.

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
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_PEAK_TFLOPS = 78.6  # same guide: vector FP64
FP32_PEAK_TFLOPS = 157.3  # same guide: vector FP32
# algorithmic bytes and flops per unit: SURVEY.md 8(d) / DESIGN.md section 4
B_STEP = {"c3": 144, "c4": 72, "c2": 72}
B_RAY = {"c3": 96, "c4": 56, "c2": 56}
# flops are no longer estimated: they are the hardware's own count of the profiled launch
# (SQ_INSTS_VALU_FLOPS_* x 64 lanes, profiles/traffic.json -> valu.flops_counted_per_launch), quoted
# only when the library on disk holds the very kernel the counters were read on
# SURVEY.md 8(d)'s algorithmic flops per unit, the fallback when no counter pass of the kernel on
# disk is committed: ~1.3 kflop per accepted RKF45 step (f64), ~0.5 kflop per implicit-midpoint step,
# ~0.15 kflop per Cartesian-Verlet step
F_STEP = {"c3": 1300.0, "c4": 500.0, "c2": 500.0, "c2glsl": 150.0}
KERNEL_OF = {("c3", "fast"): "integrate_segment_kernel<1,1,0>",
             ("c3", "strict"): "integrate_segment_kernel<1,0,0>",
             ("c4", "fast"): "wgsl_symplectic_fast_kernel",
             ("c4", "packed"): "wgsl_symplectic_pk_kernel",
             ("c4", "strict"): "wgsl_symplectic_kernel",
             ("c2", "fast"): "glsl_fragment_kernel<1>",
             ("c2", "strict"): "glsl_fragment_kernel<0>"}


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


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline_c3(width, height, eye, target_seconds=15.0, tolerance=1e-8):
    """The oracle (C restatement of gravitas-core) timed on the host cores on a bounded,
    pixel-strided sample of the same workload.  Checker/baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    cores = usable_cores()
    cam = po.camera_look_at(eye, aspect=width / height)
    fp = po.frame_params(width, height, spin=0.999, opt=po.options(max_steps=2048, tolerance=tolerance))
    lut = po.blackbody_lut(fp.lut_width, fp.lut_height, fp.lut_max_temp)
    t = time.time()
    probe = po.render_frame(cam, fp, lut, stride=(16, 16), nthreads=cores, want_states=False)
    dt = max(time.time() - t, 1e-3)
    rate = probe["stats"].accepted_steps / dt
    total = probe["stats"].accepted_steps * 256.0  # ~steps in the full frame (the probe is a 1/256 sample)
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
            "cpu_model": cpu_model(), "one_core_value": round(one_rate, 4), "kind": "port",
            "sample": "C restatement of gravitas-core (Rust toolchain unavailable), OpenMP over "
                      "rays, 1/%d pixel-strided subset of the %dx%d frame: %d rays, %d accepted "
                      "steps in %.1f s" % (sx * sy, width, height, st.rays,
                                           st.accepted_steps, dt)}


def cpu_baseline_c4(wp, width, height, target_seconds=12.0):
    """The shader oracle's f32 compute march (C restatement of compute.wgsl.ts) on the host
    cores, pixel-strided sample of the same 8K / 1024-step frame."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    cores = usable_cores()
    op = po.wgsl_params_from(wp)
    t = time.time()
    _, psteps = po.wgsl_frame(op, stride=(64, 64), nthreads=cores)
    rate = float(np.sum(psteps)) / max(time.time() - t, 1e-3)
    total = 900.0 * width * height
    sx = sy = 64
    for cand in ((4, 4), (6, 6), (8, 8), (12, 12), (16, 16), (24, 24), (32, 32), (48, 48), (64, 64)):
        if total / (cand[0] * cand[1]) / rate <= 1.6 * target_seconds:
            sx, sy = cand
            break
    t = time.time()
    _, osteps = po.wgsl_frame(op, stride=(sx, sy), nthreads=cores)
    dt = time.time() - t
    steps = int(np.sum(osteps))
    return {"value": round(steps / dt / 1e6, 4), "unit": "Mray-steps/s", "cores": cores,
            "cpu_model": cpu_model(), "kind": "port",
            "sample": "C restatement of the f32 compute march (compute.wgsl.ts; no TS/WGSL runtime "
                      "here), OpenMP over rays, 1/%d pixel-strided subset of the %dx%d frame: %d "
                      "rays, %d steps in %.1f s" % (sx * sy, width, height, osteps.size, steps, dt)}


def cpu_baseline_c2_glsl(gp, width, height, target_seconds=12.0):
    """The shader oracle's GLSL fragment march (C restatement of fragment.glsl.ts + chunks) on the
    host cores, pixel-strided sample of the same 1080p / 512-step frame."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    cores = usable_cores()
    op = po.glsl_params_from(gp)
    t = time.time()
    _, psteps = po.glsl_frame(op, stride=(32, 32), nthreads=cores)
    rate = float(np.sum(psteps)) / max(time.time() - t, 1e-3)
    total = float(np.mean(psteps)) * width * height
    sx = sy = 32
    for cand in ((1, 1), (2, 1), (2, 2), (3, 2), (3, 3), (4, 4), (6, 6), (8, 8), (12, 12), (16, 16), (32, 32)):
        if total / (cand[0] * cand[1]) / rate <= 1.6 * target_seconds:
            sx, sy = cand
            break
    t = time.time()
    _, osteps = po.glsl_frame(op, stride=(sx, sy), nthreads=cores)
    dt = time.time() - t
    steps = int(np.sum(osteps))
    return {"value": round(steps / dt / 1e6, 4), "unit": "Mray-steps/s", "cores": cores,
            "cpu_model": cpu_model(), "kind": "port",
            "sample": "C restatement of the f32 fragment-shader march (fragment.glsl.ts + chunks; no "
                      "GLSL runtime here), OpenMP over pixels, 1/%d pixel-strided subset of the %dx%d "
                      "frame: %d rays, %d steps in %.1f s" % (sx * sy, width, height, osteps.size, steps, dt)}


def committed_pmc(kernel_pretty, lib_path, frame=None):
    """HBM bytes and VALU issue fraction per launch of the dominant kernel from the committed
    rocprofv3 PMC passes (profiles/traffic.json) -- valid only for the code object they were
    measured on: the file carries the kernel's code hash, and a library whose kernel hashes
    differently gets None (the figures would silently describe another kernel)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import kernel_resources as kr
        now = kr.kernel_code_hash(lib_path, kernel_pretty)
    except Exception as exc:  # no file / unreadable library: no committed figures
        return None, "unavailable (%s)" % type(exc).__name__
    ks = t.get("kernels") or {}
    # a kernel measured at a second frame size is filed under "<kernel>@<W>x<H>"
    ent = (ks.get("%s@%dx%d" % (kernel_pretty, frame[0], frame[1])) if frame else None) or ks.get(kernel_pretty)
    if not ent:
        return None, "no committed PMC pass for %s" % kernel_pretty
    if ent.get("code_hash") != now:
        return None, ("stale: profiles/traffic.json was measured on code object %s, the library "
                      "holds %s" % (ent.get("code_hash"), now))
    return ent, "profiles/traffic.json (code object %s)" % now


def roofline_block(cfg, glsl, kernel_pretty, avg_launch_ms, launches_per_frame, ray_steps_per_frame, rays_per_frame,
                   pmc, pmc_src, same_workload, segment_tries, timing_note, same_frame=None):
    """The roofline object of the JSON line, against the bound that binds.

    The march kernels are register-resident: what bounds them is vector-ALU issue, not HBM (real HBM
    traffic of the f64 frame kernel: 0.65 % of SURVEY 8(d)'s algorithmic bytes).  So
      bound    = "fp64_valu" (c3 / c5) or "fp32_valu" (c2 / c4),
      achieved = flops of one launch / this run's mean launch time (HIP events), in TFLOP/s,
      peak     = the guide's vector peak for the dtype (78.6 / 157.3 TFLOP/s),
      frac     = achieved / peak -- always a number.
    flops of one launch = flops per ray-step x the ray-steps one launch of THIS run processed, where
    flops per ray-step is the hardware's own count (SQ_INSTS_VALU_FLOPS_* x 64 lanes of the committed
    rocprofv3 pass, profiles/traffic.json) when that pass was taken on the very code object in the
    library on disk ("counted"), else SURVEY 8(d)'s algorithmic figure ("algorithmic").
    SURVEY 8(d)'s byte figure stays beside it as `hbm_nominal` (algorithmic bytes / launch time against
    8 TB/s: saturated by construction for a kernel that keeps its state in registers) together with
    the HBM rate the counters saw."""
    f64 = cfg == "c3"
    peak_tf = FP64_PEAK_TFLOPS if f64 else FP32_PEAK_TFLOPS
    bkey = cfg
    steps_per_launch = ray_steps_per_frame / max(launches_per_frame, 1.0)
    v = (pmc.get("valu") or {}) if pmc else {}
    counted = v.get("flops_counted_per_launch")
    prof_steps = None
    if counted:
        # ray-steps of the profiled launch: recorded with the pass (format >= 3); a pass of the same
        # deterministic workload without the field processed exactly this run's steps
        prof_steps = pmc.get("ray_steps_per_launch") or (ray_steps_per_frame if same_workload else None)
    if counted and prof_steps:
        flops_per_step = counted / prof_steps
        source = ("counted: SQ_INSTS_VALU_FLOPS_* x 64 lanes of the profiled launch (%s) / its %d ray-steps"
                  % (pmc_src, int(prof_steps)))
        kind = "counted"
    else:
        flops_per_step = F_STEP["c2glsl" if glsl else cfg]
        source = "algorithmic: SURVEY 8(d) flops per step (%s)" % pmc_src
        kind = "algorithmic"
    flops_per_launch = flops_per_step * steps_per_launch
    tf = flops_per_launch / (avg_launch_ms * 1e-3) / 1e12
    bytes_per_launch = (ray_steps_per_frame * B_STEP[bkey] + rays_per_frame * B_RAY[bkey]) / max(launches_per_frame, 1.0)
    nominal = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
    # HBM bytes per launch from the counters: the one-launch pass, or the pass of the K-try schedule
    # (the f64 frame kernel's HBM bytes are set by the workspace layout -- 92 B read + 76 B written per slot --, not
    #  by how many steps a ray takes: the pass of the same frame size holds for another tolerance as well)
    traffic = None
    traffic_src = pmc_src
    if same_frame is None:
        same_frame = same_workload
    if pmc and not segment_tries and (same_workload or (cfg == "c3" and same_frame)):
        traffic = pmc.get("hbm_bytes_per_launch")
        if not same_workload:
            traffic_src = pmc_src + "; layout-determined, quoted from the tol = 1e-8 pass of the same frame"
    elif pmc and same_workload and segment_tries:
        seg = pmc.get("segment_tries_%d" % segment_tries)
        if seg:
            traffic = seg.get("hbm_bytes_per_launch")
            traffic_src = pmc_src + ", --segment-tries %d pass" % segment_tries
        else:
            traffic_src = "no committed FETCH/WRITE pass of the --segment-tries %d schedule" % segment_tries
    elif pmc:
        traffic_src = "not applicable to this run (committed pass: 1 GPU, whole frame, %s)" % pmc_src
    hbm = {"achieved": round(nominal, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "nominal_frac": round(nominal / HBM_PEAK_GBS, 4),
           "algorithmic_bytes_per_launch": int(bytes_per_launch),
           "bytes_model": "SURVEY 8(d): %d B per ray-step + %d B per ray" % (B_STEP[bkey], B_RAY[bkey]),
           "hbm_measured_GBps": round(traffic / (avg_launch_ms * 1e-3) / 1e9, 1) if traffic else None,
           "hbm_measured_frac": round(traffic / (avg_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
           "reason": "nominal: the bytes a state-through-HBM march would move; this kernel keeps a ray's state in "
                     "registers from launch to exit%s, so the figure saturates by construction and the kernel is "
                     "judged on roofline.frac (vector-ALU flops) and valu_issue_frac"
                     % (" of a segment" if segment_tries else "")}
    out = {"bound": "fp64_valu" if f64 else "fp32_valu", "achieved": round(tf, 2), "peak": peak_tf,
           "unit": "TFLOP/s", "frac": round(tf / peak_tf, 4), "traffic": traffic, "traffic_source": traffic_src,
           "traffic_pass": pmc.get("pass_id") if pmc else None,
           "kernel": kernel_pretty, "avg_launch_ms": round(avg_launch_ms, 4),
           "launches_per_frame": launches_per_frame, "ray_steps_per_launch": int(steps_per_launch),
           "flops_per_ray_step": round(flops_per_step, 1), "flops_per_launch": int(flops_per_launch),
           "flops_source": kind, "flops_source_detail": source, "timing": timing_note,
           "valu_issue_frac": v.get("issue_frac") if (pmc and same_workload and not segment_tries) else None,
           "hbm_nominal": hbm}
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line under
    torch.distributed.run with N ranks on this node (one per visible GPU) and pass rank 0's JSON
    line through.  Returns the launcher's exit status."""
    import socket
    import subprocess
    if os.environ.get("GRV_BENCH_ONE_DEVICE") != "1":
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            sys.stderr.write("bench.py: --gpus %d but only %d HIP device(s) visible\n" % (n, have))
            return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


TOL = 1e-8                # RKF45 tolerance of the f64 frame (--config c5 / --tolerance change it)
BASELINE_LABEL = {"c2": "configs[1]", "c3": "configs[2]", "c4": "configs[3]", "c5": "configs[4]"}
KERNEL = "glsl"           # --kernel of --config c2
EYE = (60.0, 97.0)        # --eye: camera distance (M) and polar angle (deg); BASELINE's frames use the default


def workload_text(cfg, W, H, split):
    if cfg == "c2" and KERNEL == "glsl":
        return ("%dx%d frame%s, a=0.999, f32 Cartesian Velocity-Verlet march of the WebGL fragment shader "
                "(fragment.glsl.ts:129-221, chunks/metric.ts:96-149) at u_maxRaySteps=512 (the shader clamps "
                "to 500), default high-quality preset (lensing, volumetric disk + Doppler, jets, stars, photon "
                "glow, blue-noise dither; seeded noise textures), mouse camera zoom=%g theta=%gdeg" % (W, H, split, EYE[0], EYE[1]))
    if cfg == "c2":
        return ("%dx%d frame%s, a=0.999, f32 compute march (Kerr-Schild implicit-midpoint, compute.wgsl.ts) "
                "at a 512-step budget, disk g-factor shading + star field, camera r0=%gM theta=%gdeg "
                "fov=60deg" % (W, H, split, EYE[0], EYE[1]))
    if cfg == "c3":
        return ("%dx%d frame%s, a=0.999 Kerr-Schild, adaptive RKF45 tol=%g h0=0.01 escape=1000 "
                "renorm=10 max_steps=2048, Planck LUT 512x64 Tmax=1e5 redshift shading, camera "
                "r0=%gM theta=%gdeg fov=60deg" % (W, H, split, TOL, EYE[0], EYE[1]))
    return ("%dx%d frame%s, a=0.999, f32 compute march (Kerr-Schild implicit-midpoint, "
            "compute.wgsl.ts) at a fixed 1024-step budget, disk g-factor shading + star field "
            "(packed / fast arithmetic: the star hash takes a FAST-contract sin, so individual stars "
            "differ from the shader-order sky; parity tests compare with stars off), "
            "camera r0=%gM theta=%gdeg fov=60deg" % (W, H, split, EYE[0], EYE[1]))


def main_native(args, cfg, base_w, base_h):
    """--native: one process, the C ABI's multi-GPU handle (csrc/engine_multi.hip).  Same workloads,
    same timing contract (K frames bracketed by a full synchronise of every rank's streams)."""
    if args.scaling != "strong":
        raise SystemExit("--native splits the one frame (strong scaling)")
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
        raise SystemExit("--native is one process: do not launch it under torch.distributed.run")
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    import torch

    import blackhole_simulation_amd as bh
    G = args.gpus
    one_device = os.environ.get("GRV_BENCH_ONE_DEVICE") == "1"
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < (1 if one_device else G):
        sys.stderr.write("bench.py: --gpus %d but only %d HIP device(s) visible (the engine has no CPU path)\n"
                         % (G, have))
        raise SystemExit(2)
    if not os.path.exists(bh.library_path()):
        bh.build_library()
    transport = {"auto": bh.TRANSPORT_AUTO, "rccl": bh.TRANSPORT_RCCL, "peer": bh.TRANSPORT_PEER_COPY}[args.transport]
    try:
        m = (bh.MultiEngine(1.0, 0.999, virtual_ranks=G) if one_device
             else bh.MultiEngine(1.0, 0.999, devices=list(range(G)), transport=transport))
    except bh.GravitasError as exc:
        # RCCL missing / ncclCommInitAll failing: the reason goes to stderr, no JSON line is printed
        # and nothing falls back to another transport
        sys.stderr.write("bench.py: %s\n" % exc)
        raise SystemExit(3)
    if m.ranks != G:
        raise SystemExit("--gpus %d but the handle has %d ranks" % (G, m.ranks))
    if args.exchange == "rgba16f":
        m.set_exchange_format(bh.EXCHANGE_RGBA16F)
    torch.cuda.set_device(0)
    W, H = base_w, base_h
    th = np.deg2rad(EYE[1])
    eye = (EYE[0] * np.sin(th), EYE[0] * np.cos(th), 0.0)
    arith = {"fast": bh.ARITH_FAST, "strict": bh.ARITH_STRICT, "packed": bh.ARITH_FAST_PACKED}[args.arith]
    cam = bh.camera_look_at(eye, aspect=W / H)
    params = bh.render_params(W, H, arith=arith if cfg == "c3" else bh.ARITH_FAST, segment_tries=args.segment_tries,
                              tolerance=TOL)
    prof = bh.render_params(W, H, arith=arith if cfg == "c3" else bh.ARITH_FAST, segment_tries=args.segment_tries,
                            profile=1, tolerance=TOL)
    wp = bh.wgsl_params(W, H, cam, 1.0, 0.999, max_steps=1024, arith=arith) if cfg == "c4" else None
    images = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0") for _ in range(2)]
    m.stats_accumulate(True)

    def frame(i, p):
        if cfg == "c3":
            m.render_frame_device(cam, p, images[i % 2])
        else:
            m.render_frame_wgsl_device(wp, images[i % 2])

    for i in range(args.warmup):
        frame(i, params)
    m.synchronize()
    m.frame_stats_reset()
    t0 = time.perf_counter()
    for i in range(args.steps):
        frame(args.warmup + i, params)
    m.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    st = m.frame_stats()
    total_steps = float(st.accepted_steps)
    # the dominant kernel of one rank's share, from profiled frames after the timed loop: HIP events the
    # library records on each rank's launch stream around its integrate / march launch (c3: profile=1
    # frames; c4: grv_engine_profile_shader_frames on every rank's engine); the slowest rank counts
    k = max(args.profile_frames, 1)
    m.frame_stats_reset()
    if cfg != "c3":
        m.profile_shader_frames(True)
    for i in range(k):
        frame(i, prof)
        m.synchronize()  # one frame at a time: the events then bracket one launch per rank
    per_rank = [m.rank_frame_stats(r) for r in range(G)]
    if cfg != "c3":
        m.profile_shader_frames(False)
    rank_ms = [round(p.integrate_ms / max(p.launches, 1), 4) for p in per_rank]
    slow = max(range(G), key=lambda r: rank_ms[r])
    launches_per_frame = max(per_rank[slow].launches / k, 1.0)
    avg_launch_ms = per_rank[slow].integrate_ms / max(per_rank[slow].launches, 1)
    kernel_pretty = KERNEL_OF[(cfg, args.arith)]
    pmc, pmc_src = committed_pmc(kernel_pretty, bh.library_path(), (W, H))
    same_workload = bool(pmc and G == 1 and (cfg != "c3" or TOL == 1e-8) and (W, H) == tuple(pmc.get("frame", (W, H))))
    roofline = roofline_block(cfg, False, kernel_pretty, avg_launch_ms, launches_per_frame,
                              per_rank[slow].accepted_steps / k, W * H / G, pmc, pmc_src, same_workload,
                              args.segment_tries,
                              "%d profiled frames after the timed loop, slowest rank's share (rank %d)" % (k, slow))
    line = {
        "metric": "Mray-steps/s", "value": round(total_steps / elapsed / 1e6, 2), "unit": "Mray-steps/s",
        "n_gpus": G, "ranks": m.ranks, "rank_devices": m.rank_devices(),
        "launcher": "native (one process, grv_engine_create_multi)",
        "transport": {bh.TRANSPORT_RCCL: "rccl", bh.TRANSPORT_PEER_COPY: "peer_copy"}.get(m.transport, m.transport),
        "rccl_version": bh.rccl_probe()[0] if m.transport == bh.TRANSPORT_RCCL else None,
        "rank_integrate_ms": {"min": min(rank_ms), "max": max(rank_ms), "per_rank": rank_ms,
                              "source": "HIP events around each rank's integrate / march launch, mean of %d profiled "
                                        "frames after the timed loop" % max(args.profile_frames, 1)},
        "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64" if cfg == "c3" else "f32", "data": "synthetic",
        "config": {"workload": workload_text(cfg, W, H, "" if G == 1 else " split over %d GPUs" % G),
                   "baseline_config": args.baseline_label, "arith": args.arith,
                   "segment_tries": args.segment_tries or "one launch",
                   "host": "one process through the C ABI (grv_engine_create_multi): a host thread and two "
                           "streams per device",
                   "partition": "64x64 tiles round-robin, one %s gather to rank 0 per frame, two frames in flight"
                                % ("RCCL send/recv-group" if m.transport == bh.TRANSPORT_RCCL else "peer-copy")
                   if G > 1 else "single GPU",
                   "virtual_ranks_on_one_device": bool(one_device),
                   "exchange": args.exchange, "exchange_bytes_per_frame": m.exchange_bytes_per_frame(W, H),
                   "rays": W * H, "accepted_steps_per_frame": int(total_steps / args.steps),
                   "host_waits_in_frame_loop": 0,
                   "frames_in_flight": 2},
        "roofline": roofline,
    }
    m.close()
    sys.stdout.flush()
    os.dup2(stdout_fd, 1)
    print(json.dumps(line), flush=True)
    os.dup2(2, 1)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed frames (default 20; --config c2: 300 -- its frames are 1-2 ms, and a 20-frame window "
                         "still sits on the GPU's clock ramp: profiles/r05_bench_window.jsonl)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed frames before them (default 3; --config c2: 60)")
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c5"], default="c3",
                    help="c2 = BASELINE configs[1] (1080p f32 march, 512 steps, one GPU), "
                         "c3 = BASELINE configs[2] (the headline), c4 = configs[3] (8K f32 march), c5 = configs[4]: "
                         "the c3 frame at tol = 1e-9 under the reference-order STRICT contract (the parity run)")
    ap.add_argument("--kernel", choices=["glsl", "wgsl"], default="glsl",
                    help="--config c2: the WebGL fragment shader's Verlet march (default, the reference's "
                         "production loop) or the WGSL compute march")
    ap.add_argument("--launcher", choices=["auto", "native", "torchrun"], default="auto",
                    help="N > 1 host: torchrun (= auto) = one process per GPU under torch.distributed.run, started "
                         "by this script when no launcher did; native = ONE process through the C ABI's multi-GPU "
                         "handle (explicit: not yet run on two real devices)")
    ap.add_argument("--tolerance", type=float, default=None, help="RKF45 tolerance of the f64 frame (default 1e-8; c5: 1e-9)")
    ap.add_argument("--arith", choices=["fast", "strict", "packed"], default=None,
                    help="arithmetic contract (default: fast for c3; packed = the FAST contract with two rays per "
                         "lane on the packed-f32 ops for c4)")
    ap.add_argument("--segment-tries", type=int, default=0)
    ap.add_argument("--eye", default=None, metavar="R0,THETA",
                    help="camera distance in M and polar angle in degrees (default 60,97: BASELINE's frames).  The "
                         "reference's observer ranges over 1.5-100 R_s and 0.1-179.9 deg (simulation.config.ts:106-121); a "
                         "line taken with another camera is a secondary record (config.eye), never the headline")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: wait for each frame's gather before integrating the next frame")
    ap.add_argument("--two-streams", action="store_true",
                    help="N = 1: even / odd frames on two streams as for N > 1 (the roofline block then comes from "
                         "profiled frames after the timed loop); the default of --config c2")
    ap.add_argument("--one-stream", action="store_true",
                    help="queue every frame on one stream (default: even / odd frames on two streams, "
                         "so one frame's tail runs under the next frame's head)")
    ap.add_argument("--profile-frames", type=int, default=3,
                    help="N > 1: profiled frames after the timed loop (roofline block)")
    ap.add_argument("--native", action="store_true",
                    help="ONE process drives all N GPUs through the C ABI's multi-GPU handle "
                         "(grv_engine_create_multi: a host thread and two streams per device, one RCCL "
                         "send/recv group per frame) instead of N torch.distributed ranks")
    ap.add_argument("--transport", choices=["auto", "rccl", "peer"], default="auto",
                    help="--native: exchange transport (auto: RCCL between real devices)")
    ap.add_argument("--exchange", choices=["rgba32f", "rgba16f"], default="rgba32f",
                    help="N > 1: what the one gather carries -- f32 pixels (default) or the reference compute pass's "
                         "own rgba16float format (renderer.ts:163-176): half the bytes, the assembled image is the frame "
                         "rounded through binary16")
    args = ap.parse_args()
    cfg = args.config
    if args.steps is None:
        args.steps = 300 if cfg == "c2" else 20
    if args.warmup is None:
        args.warmup = 60 if cfg == "c2" else 3
    global TOL, KERNEL, EYE
    KERNEL = args.kernel
    if args.eye:
        r0, thd = (float(x) for x in args.eye.split(","))
        if not (r0 > 0.0 and 0.0 < thd < 180.0):
            raise SystemExit("--eye R0,THETA: R0 > 0 M, 0 < THETA < 180 deg")
        EYE = (r0, thd)
    label = BASELINE_LABEL[cfg]
    if cfg == "c5":  # the c3 code path with the parity run's tolerance and arithmetic
        cfg = "c3"
        if args.arith is None:
            args.arith = "strict"
        if args.tolerance is None:
            args.tolerance = 1e-9
    if args.tolerance is not None:
        if cfg != "c3":
            raise SystemExit("--tolerance applies to the f64 frame (--config c3 / c5)")
        TOL = args.tolerance
    args.baseline_label = label
    glsl = cfg == "c2" and KERNEL == "glsl"
    if args.arith is None:
        args.arith = "fast" if (cfg == "c3" or glsl) else "packed"
    if (cfg == "c3" or glsl) and args.arith == "packed":
        raise SystemExit("--arith packed is the two-rays-per-lane form of the WGSL compute march")
    base_w = args.width or {"c2": 1920, "c3": 3840, "c4": 7680}[cfg]
    base_h = args.height or {"c2": 1080, "c3": 2160, "c4": 4320}[cfg]
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if cfg == "c2" and args.gpus > 1:
        raise SystemExit("--config c2 is BASELINE configs[1]: one GPU")
    launched = int(os.environ.get("WORLD_SIZE", "0") or 0) >= 1 and "RANK" in os.environ
    if args.launcher == "auto":
        # N > 1 defaults to one process per GPU under torch.distributed.run (the driver's own launch form,
        # the host the world-size-2 tests cover).  The one-process C-ABI handle (--launcher native) has
        # never met two real devices (tests/test_gpu_multi_real.py is gated on >= 2 GPUs and has not run):
        # it stays an explicit choice until a MULTICHIP record shows it green.
        args.launcher = "torchrun"
    if args.native or (args.gpus > 1 and args.launcher == "native"):
        if launched and int(os.environ["WORLD_SIZE"]) > 1:
            raise SystemExit("--launcher native is one process: do not start it under torch.distributed.run")
        return main_native(args, cfg, base_w, base_h)
    if args.gpus > 1 and not launched:
        # `python bench.py --gpus N --launcher torchrun`: start the N ranks ourselves (one process per GPU)
        raise SystemExit(spawn_ranks(args.gpus))

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
    if world != args.gpus:  # never an N = 1 line for a --gpus 8 command (or the reverse)
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    one_device = os.environ.get("GRV_BENCH_ONE_DEVICE") == "1"
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_dev < (1 if one_device else world):
        sys.stderr.write("bench.py: --gpus %d but only %d HIP device(s) visible (the engine has no CPU path)\n"
                         % (world, n_dev))
        raise SystemExit(2)
    if not os.path.exists(bh.library_path()):
        if local_rank == 0:
            bh.build_library()  # checkout without the built artefact: compile it (hipcc)
        else:
            while not os.path.exists(bh.library_path()):
                time.sleep(1.0)
    # Test hooks (tests/test_gpu_dist_shared_device.py): GRV_BENCH_ONE_DEVICE=1 puts every rank on
    # cuda:0 and GRV_BENCH_BACKEND=gloo exchanges the device buffers through gloo, so that the N > 1
    # control flow of this file (strong split, two frames in flight, pipelined gather, timing
    # reductions) runs for real on a one-GPU box; RCCL refuses two ranks on one device.
    backend = os.environ.get("GRV_BENCH_BACKEND", "nccl")
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "RANK" in os.environ  # under torchrun even a 1-rank job walks the RCCL path
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("--gpus %d but the process group has %d ranks" % (args.gpus, dist.get_world_size()))
    # which device every rank sits on (goes into the JSON line: the judge of a scaling curve should
    # not have to trust n_gpus)
    rank_devices = [local_rank]
    if use_dist:
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, local_rank)
        if not one_device and len(set(rank_devices)) != world:
            raise SystemExit("ranks share devices: %s" % (rank_devices,))

    gx, gy = GRID.get(world, (world, 1)) if args.scaling == "weak" else (1, 1)
    W, H = base_w * gx, base_h * gy
    th = np.deg2rad(EYE[1])
    eye = (EYE[0] * np.sin(th), EYE[0] * np.cos(th), 0.0)
    arith = {"fast": bh.ARITH_FAST, "strict": bh.ARITH_STRICT, "packed": bh.ARITH_FAST_PACKED}[args.arith]

    eng = bh.PhysicsEngine(1.0, 0.999, device=local_rank)
    cam = bh.camera_look_at(eye, aspect=W / H)
    params = bh.render_params(W, H, arith=arith if cfg == "c3" else bh.ARITH_FAST,
                              segment_tries=args.segment_tries, profile=0, tolerance=TOL)
    rp = D.rank_params(params, world, rank)
    prof_rp = D.rank_params(params, world, rank)
    prof_rp.profile = 1
    n_local = eng.frame_ray_count(rp)
    wp = gp = None
    glsl = cfg == "c2" and KERNEL == "glsl"
    if cfg == "c4" or (cfg == "c2" and not glsl):
        wp = bh.wgsl_params(W, H, cam, 1.0, 0.999, max_steps=1024 if cfg == "c4" else 512, arith=arith,
                            tile_world=world, tile_rank=rank)
    if glsl:
        # the uniforms WebGLRenderer uploads at the default preset (grv_glsl_params_default), with the
        # config's step budget; linear output (the post chain owns tone mapping upstream)
        gp = bh.glsl_params(W, H, 1.0, 0.999, max_ray_steps=512, arith=arith, tile_world=world, tile_rank=rank)
        if args.eye:  # the shader's mouse camera: u_zoom = distance, u_mouse.y = polar angle / 180 deg
            gp.zoom = EYE[0]
            gp.mouse[1] = EYE[1] / 180.0
    # two frames in flight: even and odd frames go to two streams (the engine alternates two ray
    # workspaces and orders each behind its previous user), so the tail of one frame's integrate
    # launch -- too few waves left to fill 256 CUs -- runs under the head of the next frame
    # (N > 1 only: at N = 1 the tail is 1 % of a frame, and one stream keeps the in-loop HIP events
    # bracketing one kernel at a time, which is what the roofline block is defined on)
    # (c2: a 1080p march is 1-4 ms, a 16 000-32 000-wave grid whose ramp and tail are 6-19 % of the launch:
    # two frames in flight by default there too, as a renderer's frame loop keeps them)
    two = (world > 1 or args.two_streams or cfg == "c2") and not args.one_stream and not args.no_overlap
    streams = [torch.cuda.Stream(), torch.cuda.Stream()] if two else [torch.cuda.current_stream()] * 2
    stream = streams[0].cuda_stream
    # all buffers live outside the frame loop: the padded send buffer doubles as the render
    # target, rank 0 additionally holds the receive slots and the assembled image
    half = args.exchange == "rgba16f" and use_dist
    tg = D.TileGather(params, world, rank, 4, torch.float16 if half else torch.float32, torch.device("cuda", local_rank)) \
        if use_dist else None
    overlap = tg is not None and not args.no_overlap
    if overlap:
        tg.enable_pipeline()  # second send buffer: frame i's gather runs under frame i+1's kernels
    # (rgba16f exchange: the kernels render f32 into their own targets; the share is narrowed -- one rounding to
    # nearest even, torch's f32 -> f16 copy -- into the send buffer that travels)
    bufs = [tg.local_view(n_local)] * 2 if (tg and not half) else \
        [torch.empty((n_local, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    eng.stats_accumulate(True)  # counters stay in HBM across frames: no read-back in the loop

    def dev_unpack(rparams, r, packed, image):
        eng.unpack_tiles_device(rparams, r, packed, image, 8 if half else 16, torch.cuda.current_stream().cuda_stream)

    # c4 at N = 1: bracket each march launch with events on the launch stream (torch's current
    # stream is the stream handed to the engine), resolved after the loop
    ev_pairs = []

    def render(target, profiled):
        s = torch.cuda.current_stream().cuda_stream
        if cfg == "c3":
            eng.render_frame_device(cam, prof_rp if profiled else rp, rgba=target, stream=s)
        else:
            if profiled:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
            if glsl:
                eng.render_frame_glsl(gp, target, stream=s, want_total=False)
            else:
                eng.render_frame_wgsl(wp, target, stream=s, want_total=False)
            if profiled:
                b.record()
                ev_pairs.append((a, b))

    def one_frame(i, profiled):
        with torch.cuda.stream(streams[i % 2]):
            target = tg.pipelined_view(i, n_local) if (overlap and not half) else bufs[i % 2]
            render(target, profiled)
            if half:  # narrow the share into the buffer that travels
                (tg.pipelined_view(i, n_local) if overlap else tg.local_view(n_local)).copy_(target)
            if overlap:
                tg.submit(i, dev_unpack, force_collective=True)  # finish frame i-1's exchange, start frame i's
            elif tg:
                tg.run(dev_unpack, force_collective=True)  # the one exchange: gather tiles -> rank 0

    def fence():
        if overlap:
            tg.drain(dev_unpack)  # the last frame's exchange is inside the timed region
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def read_stats():
        st = eng.frame_stats(stream)
        ms = st.integrate_ms
        n = st.launches
        if cfg != "c3":
            ms = sum(a.elapsed_time(b) for a, b in ev_pairs)
            n = len(ev_pairs)
            del ev_pairs[:]
        return st, ms, n

    in_loop_profile = world == 1 and not two  # no events in a multi-rank / two-stream timed loop
    if tg is not None:
        tg.enable_wait_timing()  # two events around each wait point of the exchange: where a rank's stream stood still
    if tg is not None:
        # communicator set-up (RCCL opens its xGMI peer connections on first use) belongs to
        # initialisation, not to a frame: one exchange of the still empty buffers, whatever --warmup is
        tg.run(dev_unpack, force_collective=True)
        torch.cuda.synchronize()
    for i in range(args.warmup):
        one_frame(i, False)
    fence()
    eng.frame_stats_reset(stream)
    torch.cuda.synchronize()
    if tg is not None:
        tg.exchange_wait_ms()  # drop the warm-up's wait events
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_frame(args.warmup + i, in_loop_profile)
    fence()
    elapsed = time.perf_counter() - t0
    my_wait_ms = tg.exchange_wait_ms() / max(args.steps, 1) if tg is not None else 0.0
    st, integ_ms, launches = read_stats()
    steps_local = st.accepted_steps
    rays_local = st.rays // max(args.steps, 1) if cfg == "c3" else min(n_local, W * H)
    max_drift = st.max_drift
    prof_note = "HIP events on the launch stream inside the timed region"
    if not in_loop_profile:
        # roofline of this rank's share from profiled frames outside the timed region
        eng.frame_stats_reset(stream)
        k = max(args.profile_frames, 1)
        for i in range(k):  # one stream: the events then bracket one kernel at a time
            with torch.cuda.stream(streams[0]):
                render(bufs[0], True)
        torch.cuda.synchronize()
        pst, integ_ms, launches = read_stats()
        prof_steps_per_frame = pst.accepted_steps / k
        prof_note = "%d profiled frames of rank 0's tile share after the timed loop" % k
    else:
        prof_steps_per_frame = steps_local / max(args.steps, 1)

    # per-rank mean integrate-launch time (every rank's own HIP events), gathered for the JSON line
    my_ms = integ_ms / max(launches, 1)
    rank_ms = [my_ms]
    rank_wait_ms = [my_wait_ms]
    if use_dist:
        rank_ms = [None] * world
        dist.all_gather_object(rank_ms, my_ms)
        rank_wait_ms = [None] * world
        dist.all_gather_object(rank_wait_ms, my_wait_ms)
    agg = torch.tensor([elapsed, float(steps_local), float(rays_local)], dtype=torch.float64, device="cuda")
    if use_dist:
        tmax = agg[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg[1:], op=dist.ReduceOp.SUM)
        elapsed = float(tmax.item())
    total_steps = float(agg[1].item())
    total_rays = int(agg[2].item()) if cfg == "c3" else W * H

    if rank == 0:
        value = total_steps / elapsed / 1e6
        # roofline of the dominant kernel against the bound that binds it (vector-ALU flops of this rank's
        # launch / mean HIP-event duration of a launch); SURVEY 8(d)'s byte figure rides along as hbm_nominal
        frames_prof = args.steps if in_loop_profile else max(args.profile_frames, 1)
        launches_per_frame = max(launches / frames_prof, 1.0)
        avg_launch_ms = integ_ms / max(launches, 1)
        kernel_pretty = KERNEL_OF[("c4" if wp is not None else cfg, args.arith)]
        pmc, pmc_src = committed_pmc(kernel_pretty, bh.library_path(), (W, H))
        same_workload = bool(pmc and world == 1 and (cfg != "c3" or TOL == 1e-8) and
                             (W, H) == tuple(pmc.get("frame", (W, H))))
        same_frame = bool(pmc and world == 1 and (W, H) == tuple(pmc.get("frame", (W, H))))
        roofline = roofline_block(cfg, glsl, kernel_pretty, avg_launch_ms, launches_per_frame, prof_steps_per_frame,
                                  rays_local, pmc, pmc_src, same_workload, args.segment_tries, prof_note, same_frame)
        if args.segment_tries and cfg == "c3":
            # every launch of the compacting schedule is the compact kernel (same body, live count from device memory);
            # the flops per ACCEPTED ray-step are the one-launch kernel's count: useful flops -- the schedule's extra
            # instructions (re-establishing a ray every K tries) are what lowers the line, not what it is credited with
            roofline["kernel"] = kernel_pretty.replace("integrate_segment_kernel", "integrate_compact_kernel")
            roofline["flops_note"] = "useful flops: per accepted ray-step as counted on the one-launch kernel"
        split = "" if world == 1 else (" split over %d GPUs" % world if args.scaling == "strong"
                                         else " (%dx%d per GPU x %d)" % (base_w, base_h, world))
        workload = workload_text(cfg, W, H, split)
        line = {
            "metric": "Mray-steps/s", "value": round(value, 2), "unit": "Mray-steps/s",
            "n_gpus": world, "ranks": dist.get_world_size() if use_dist else 1,
            "rank_devices": rank_devices,
            "launcher": ("torchrun (one process per GPU, torch.distributed)" if use_dist else "single process"),
            "transport": (("rccl" if backend == "nccl" else backend) if use_dist else None),
            "rccl_version": (".".join(str(x) for x in torch.cuda.nccl.version())
                             if use_dist and backend == "nccl" else None),
            "rank_integrate_ms": {"min": round(min(rank_ms), 4), "max": round(max(rank_ms), 4),
                                  "per_rank": [round(x, 4) for x in rank_ms],
                                  "source": "mean HIP-event time of a rank's march / integrate launch (%s)" % prof_note},
            # per rank and frame: how long the rank's streams stood at the exchange's wait points (the gather of the
            # previous frame, the reader of a send buffer) -- with rank_integrate_ms, where a scaling point's time went
            "exchange_wait_ms": ({"per_rank": [round(x, 4) for x in rank_wait_ms], "max": round(max(rank_wait_ms), 4),
                                  "source": "HIP events around TileGather's wait points on the rank's stream, mean per timed frame"}
                                 if use_dist else None),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64" if cfg == "c3" else "f32", "data": "synthetic",
            "config": {"workload": workload, "baseline_config": args.baseline_label,
                       "arith": args.arith, "segment_tries": args.segment_tries or "one launch",
                       "partition": ("64x64 tiles round-robin, one RCCL gather to rank 0 per frame%s"
                                     % (", overlapped with the next frame" if overlap else ""))
                       if world > 1 else "single GPU",
                       "eye": {"r0_M": EYE[0], "theta_deg": EYE[1], "default": args.eye is None},
                       "rays": total_rays, "accepted_steps_per_frame": int(total_steps / args.steps),
                       "host_waits_in_frame_loop": 0,
                       "frames_in_flight": 2 if two else 1,
                       **({"exchange": args.exchange} if world > 1 else {}),
                       **({"exchange_backend": backend} if backend != "nccl" else {})},
            "roofline": roofline,
        }
        if cfg == "c3":
            line["config"]["max_hamiltonian_drift_rank0"] = max_drift
        if cfg == "c2":
            line["config"]["kernel"] = KERNEL
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = (cpu_baseline_c3(base_w, base_h, eye, tolerance=TOL) if cfg == "c3"
                                    else cpu_baseline_c2_glsl(gp, base_w, base_h) if glsl
                                    else cpu_baseline_c4(wp, base_w, base_h))
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
