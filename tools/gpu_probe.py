"""Exploratory GPU run: parity spot checks vs the oracle + 4K timing sweep."""
import sys, os, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import blackhole_simulation_amd as bh
import pyoracle as po
import torch

def cmp_states(a, b):
    d = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    return d.max(axis=1)

def batch_check():
    rng = np.random.default_rng(1)
    n = 4096
    st = np.zeros((n, 8))
    st[:, 1] = rng.uniform(5, 60, n); st[:, 2] = rng.uniform(0.3, 2.8, n); st[:, 3] = rng.uniform(0, 6, n)
    st[:, 4] = -1.0; st[:, 5] = rng.uniform(-1, 0.2, n); st[:, 6] = rng.uniform(-6, 6, n); st[:, 7] = rng.uniform(-6, 6, n)
    eng = bh.PhysicsEngine(1.0, 0.9)
    for kind in (bh.KERR_KS, bh.KERR_BL, bh.SCHWARZSCHILD):
        m = po.metric(kind, 1.0, 0.9)
        ref = po.integrate_batch(m, po.options(max_steps=2000), st, nthreads=8)
        for arith in (bh.ARITH_STRICT, bh.ARITH_FAST):
            o = bh.engine.default_options(metric_kind=kind, max_steps=2000, arith=arith)
            t = time.time(); got = eng.integrate_batch(st, o); dt = time.time() - t
            e = cmp_states(got["states"], ref["states"])
            same_t = (got["term"] == ref["term"]).mean()
            ds = np.abs(got["steps"].astype(int) - ref["steps"].astype(int))
            print(f"batch kind={kind} arith={arith}: term_match={same_t:.4f} steps_eq={(ds==0).mean():.4f} max_dsteps={ds.max()} "
                  f"err p50={np.median(e):.2e} p99={np.percentile(e,99):.2e} max={e.max():.2e} drift_max_rel={np.max(np.abs(got['drift']-ref['drift'])):.2e} t={dt:.3f}s")
    # single ray entry
    v = [0, 20, np.pi/2, 0, -1, -1, 0, 3.5]
    a = eng.integrate_ray_relativistic(v, 10000, 1e-8, True); b = po.integrate_ray_relativistic(1.0, 0.9, v, 10000, 1e-8, True)
    print("single KS", a, np.abs(a-b).max())
    a = eng.integrate_ray_relativistic(v, 10000, 1e-8, False); b = po.integrate_ray_relativistic(1.0, 0.9, v, 10000, 1e-8, False)
    print("single BL", np.abs(a-b).max(), "short echo", eng.integrate_ray_relativistic([1,2,3], 10, 1e-8, True))
    eng.close()

def frame_check():
    W, H = 256, 144
    th = np.deg2rad(97.0); r0 = 60.0
    eye = (r0*np.sin(th), r0*np.cos(th), 0.0)
    eng = bh.PhysicsEngine(1.0, 0.999)
    cam = bh.camera_look_at(eye, aspect=W/H)
    ocam = po.camera_look_at(eye, aspect=W/H)
    lut_o = po.blackbody_lut(512, 64, 1e5)
    lut_g = eng.generate_spectrum_lut(512, 64, 1e5)
    rel = np.abs(lut_g - lut_o) / np.maximum(np.abs(lut_o), 1e-30)
    print("LUT max rel diff", rel.max(), "exact frac", (lut_g == lut_o).mean())
    ref = po.render_frame(ocam, po.frame_params(W, H), lut_o, nthreads=8)
    for arith in (0, 1):
        p = bh.render_params(W, H, arith=arith)
        n = W*H
        dev = torch.device("cuda:0")
        rgba = torch.zeros(n, 4, dtype=torch.float32, device=dev); fs = torch.zeros(n, 8, dtype=torch.float64, device=dev)
        steps = torch.zeros(n, dtype=torch.int32, device=dev); term = torch.zeros(n, dtype=torch.uint8, device=dev); drift = torch.zeros(n, dtype=torch.float64, device=dev)
        eng.render_frame_device(cam, p, rgba, fs, steps, term, drift)
        st = eng.frame_stats()
        torch.cuda.synchronize()
        e = cmp_states(fs.cpu().numpy(), ref["states"])
        tm = (term.cpu().numpy() == ref["term"]).mean()
        ds = np.abs(steps.cpu().numpy().astype(int) - ref["steps"].astype(int))
        cd = np.abs(rgba.cpu().numpy().reshape(H, W, 4) - ref["rgba"]).max() / max(ref["rgba"][..., :3].max(), 1e-30)
        print(f"frame arith={arith}: steps {st.accepted_steps} vs {ref['stats'].accepted_steps} tries {st.rkf_tries} vs {ref['stats'].rkf_tries} terms {list(st.term_count)} vs {list(ref['stats'].term_count)} cross {st.crossings} vs {ref['stats'].crossings}")
        print(f"   term_match={tm:.5f} steps_eq={(ds==0).mean():.5f} max_dsteps={ds.max()} err p50={np.median(e):.2e} p99={np.percentile(e,99):.2e} p999={np.percentile(e,99.9):.2e} max={e.max():.2e} rgba_maxdiff_rel={cd:.2e} launches={st.launches}")
    eng.close()

def timing():
    W, H = 3840, 2160
    th = np.deg2rad(97.0); r0 = 60.0
    eye = (r0*np.sin(th), r0*np.cos(th), 0.0)
    eng = bh.PhysicsEngine(1.0, 0.999)
    cam = bh.camera_look_at(eye, aspect=W/H)
    n = W*H
    dev = torch.device("cuda:0")
    rgba = torch.zeros(n, 4, dtype=torch.float32, device=dev)
    steps = torch.zeros(n, dtype=torch.int32, device=dev)
    p = bh.render_params(W, H, arith=1, segment_tries=4096)
    eng.render_frame_device(cam, p, rgba, None, steps); torch.cuda.synchronize()
    st2 = steps.view(H // 8, 8, W // 8, 8).permute(0, 2, 1, 3).reshape(-1, 64).double()
    print("wave(8x8) efficiency mean/max:", (st2.sum() / (st2.max(dim=1).values.sum() * 64)).item(),
          " steps mean", st2.mean().item(), "max", st2.max().item())
    for arith in (1, 0):
        for K in (8, 16, 32, 64, 4096):
            p = bh.render_params(W, H, arith=arith, segment_tries=K, profile=1)
            eng.render_frame_device(cam, p, rgba); st = eng.frame_stats()  # warm
            torch.cuda.synchronize(); t = time.time()
            eng.render_frame_device(cam, p, rgba); st = eng.frame_stats()
            torch.cuda.synchronize(); dt = time.time() - t
            print(f"4K arith={arith} K={K}: wall {dt*1e3:.1f} ms  steps {st.accepted_steps/1e6:.1f}M tries {st.rkf_tries/1e6:.1f}M launches {st.launches} "
                  f"init {st.init_ms:.2f} integ {st.integrate_ms:.2f} shade {st.shade_ms:.2f} total {st.total_ms:.2f} ms -> {st.accepted_steps/dt/1e6:.0f} Mray-steps/s  frac={(st.accepted_steps*144+n*96)/dt/8e12:.3f}", flush=True)
    eng.close()

def shader_timing():
    eng = bh.PhysicsEngine(1.0, 0.999)
    for (W, H, ms) in ((1920, 1080, 512), (3840, 2160, 512), (7680, 4320, 1024)):
        n = W * H
        rgba = torch.zeros(n, 4, dtype=torch.float32, device="cuda:0")
        cam = bh.camera_look_at(EYE3, aspect=W / H)
        for name, fn, gp in (("wgsl-symplectic-f32", eng.render_frame_wgsl, bh.wgsl_params(W, H, cam, 1.0, 0.999, max_steps=ms)),
                             ("glsl-verlet-f32", eng.render_frame_glsl, bh.glsl_params(W, H, 1.0, 0.999, max_ray_steps=ms))):
            fn(gp, rgba); torch.cuda.synchronize()
            t = time.time(); reps = 5
            for _ in range(reps):
                tot = fn(gp, rgba)
            torch.cuda.synchronize(); dt = (time.time() - t) / reps
            print(f"{name} {W}x{H} max_steps={ms}: {dt*1e3:.2f} ms/frame, {tot/1e6:.1f} M steps, {tot/dt/1e6:.0f} Mray-steps/s, "
                  f"frac(72 B/step + 16 B/ray)={(tot*72+n*16)/dt/8e12:.3f}", flush=True)
    eng.close()

EYE3 = (60.0 * np.sin(np.deg2rad(97.0)), 60.0 * np.cos(np.deg2rad(97.0)), 0.0)

if __name__ == "__main__":
    what = sys.argv[1:] or ["batch", "frame", "timing"]
    if "batch" in what: batch_check()
    if "frame" in what: frame_check()
    if "timing" in what: timing()
    if "shader" in what: shader_timing()
