"""diagnostic (GPU box): component-wise FAST - oracle differences of the worst matched rays of a few fuzz configurations"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import blackhole_simulation_amd as bh, pyoracle as po
import test_fuzz_parity as T

def comp(a, b, steps_a, ref, tag):
    ok = np.isfinite(a).all(1) & np.isfinite(b).all(1)
    same = ok & (steps_a.astype(np.int64) == ref["steps"].astype(np.int64))
    rel = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    err = rel.max(1)
    idx = np.flatnonzero(same & (err > 1e-5))
    worst = idx[np.argsort(err[idx])[-8:]]
    out = dict(tag=tag, matched=int(same.sum()), above=int(idx.size),
               worst_component_hist=np.bincount(rel[idx].argmax(1), minlength=8).tolist() if idx.size else [],
               term_hist_above=np.bincount(ref["term"][idx], minlength=5).tolist() if idx.size else [],
               term_hist_all=np.bincount(ref["term"][same], minlength=5).tolist())
    # Cartesian position error of the same rays
    def cart(s):
        r, th, ph = s[:, 1], s[:, 2], s[:, 3]
        return np.stack([r * np.sin(th) * np.cos(ph), r * np.cos(th), r * np.sin(th) * np.sin(ph)], 1)
    ce = np.abs(cart(a) - cart(b)).max(1) / np.maximum(1.0, np.abs(b[:, 1]))
    out["cart_err_max_matched"] = float(ce[same].max(initial=0.0))
    out["cart_above_1e5"] = int((ce[same] > 1e-5).sum())
    out["t_pr_err_max"] = float(np.maximum(rel[same][:, 0], rel[same][:, 5]).max(initial=0.0))
    out["rays"] = [dict(i=int(i), rel=[float("%.2e" % x) for x in rel[i]], ref=[float("%.6g" % x) for x in b[i]],
                        term=int(ref["term"][i]), steps=int(ref["steps"][i]), cart=float(ce[i])) for i in worst]
    print(json.dumps(out))

# batch: BL extremal (seed 22-like)
rng = np.random.default_rng(7)
for kind, okind, spin, mass, tol, esc in ((bh.KERR_BL, po.KERR_BL, -1.0, 1.0, 4.96e-10, 80.0), (bh.KERR_BL, po.KERR_BL, 1.0, 0.37, 6.4e-8, 1000.0),
                                          (bh.KERR_BL, po.KERR_BL, 0.9, 1.0, 1e-8, 1000.0), (bh.KERR_KS, po.KERR_KS, 1.0, 1.0, 1e-8, 1000.0)):
    st = T._well_posed(rng, 3000, mass)
    kw = dict(method=0, tolerance=tol, initial_step=0.5, max_steps=2048, escape_radius=esc, renormalize_interval=10)
    m = po.metric(okind, mass, spin)
    ref = po.integrate_batch(m, po.options(**kw), st, nthreads=8)
    with bh.PhysicsEngine(mass, spin) as e:
        got = e.integrate_batch(st, bh.engine.default_options(metric_kind=kind, arith=bh.ARITH_FAST, **kw))
    comp(got["states"], ref["states"], got["steps"], ref, dict(case="batch", kind=int(okind), spin=spin, tol=tol))
# frames: KS camera on the axis (seed 10 / 18 / 12-like)
for spin, r0, th, fovy, tol, ms in ((0.5, 300.0, 1e-6, 20.0, 2.18e-10, 600), (1.0, 60.0, 1e-6, 110.0, 3e-10, 2048), (0.999, 8.0, 1e-6, 110.0, 4.6e-10, 2048),
                                    (0.999, 60.0, 0.0, 60.0, 1e-8, 2048), (0.999, 60.0, 0.3, 60.0, 1e-8, 2048)):
    W, H = 96, 72
    eye = (r0 * np.sin(th), r0 * np.cos(th), 0.0)
    up = (0.0, 1.0, 0.0) if th not in (0.0, np.pi) else (1.0, 0.0, 0.0)
    okw = dict(max_steps=ms, tolerance=tol, escape_radius=1000.0, renormalize_interval=1, method=0, initial_step=1.0)
    ref = po.render_frame(po.camera_look_at(eye, up=up, fovy_deg=fovy, aspect=W / H), po.frame_params(W, H, spin=spin, opt=po.options(**okw)), None, nthreads=8)
    n = W * H
    with bh.PhysicsEngine(1.0, spin) as e:
        cam = bh.camera_look_at(eye, up=up, fovy_deg=fovy, aspect=W / H)
        p = bh.render_params(W, H, arith=bh.ARITH_FAST, **okw)
        fs = torch.zeros(n, 8, dtype=torch.float64, device="cuda:0"); steps = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        e.render_frame_device(cam, p, None, fs, steps)
        torch.cuda.synchronize()
    comp(fs.cpu().numpy(), ref["states"], steps.cpu().numpy().astype(np.uint32), ref, dict(case="frame", spin=spin, r0=r0, theta=th, fovy=fovy, tol=tol))
