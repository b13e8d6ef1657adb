"""The integrate schedules of the f64 frame are orderings of the same per-ray arithmetic: whatever order the
waves start in, however often the live rays are re-listed, the frame must come out bit for bit the same
(north_star: "wavefront ballots for adaptive-step early-out / ray compaction").

 * GRV_SCHEDULE_DEFAULT: one launch, one-wave blocks dispatched longest-first by the PREVIOUS frame's per-wave
   tries (finalize kernel -> counting sort on a side stream -> SegmentParams.order);
 * GRV_SCHEDULE_SLOT_ORDER: one launch in slot order;
 * segment_tries = K: the compacting schedule."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 1920, 1080   # 2 073 600 rays (whole frames of 65 536 rays and more take a measured order)


def _frame(bh, eng, eye, stream=None, **kw):
    import torch
    n = W * H
    out = dict(rgba=torch.full((n, 4), -1.0, dtype=torch.float32, device="cuda"),
               final_state=torch.zeros(n, 8, dtype=torch.float64, device="cuda"),
               steps=torch.zeros(n, dtype=torch.int32, device="cuda"),
               termination=torch.zeros(n, dtype=torch.uint8, device="cuda"),
               drift=torch.zeros(n, dtype=torch.float64, device="cuda"))
    p = bh.render_params(W, H, arith=kw.pop("arith", bh.ARITH_FAST), tolerance=1e-7, max_steps=600, **kw)
    eng.render_frame_device(bh.camera_look_at(eye, aspect=W / H), p, stream=stream, **out)
    return out


def _bits(out):
    import torch
    torch.cuda.synchronize()
    return (out["rgba"].cpu().numpy().view(np.uint32), out["final_state"].cpu().numpy().view(np.uint64),
            out["steps"].cpu().numpy(), out["termination"].cpu().numpy(), out["drift"].cpu().numpy().view(np.uint64))


def _eye(r0, th_deg):
    th = np.deg2rad(th_deg)
    return (r0 * np.sin(th), r0 * np.cos(th), 0.0)


@pytest.mark.parametrize("arith", [1, 0])
def test_measured_wave_order_never_changes_a_frame(engine_mod, arith):
    """Frame 1 of a geometry runs in slot order, frames 2.. in the order the frame before last produced (the
    cameras move, so every order is stale by construction), on one stream and alternating two."""
    import torch
    bh = engine_mod
    cams = [_eye(20.0, 97.0), _eye(20.0, 97.0), _eye(12.0, 80.0), _eye(20.0, 97.0), _eye(6.0, 91.0), _eye(20.0, 97.0)]
    with bh.PhysicsEngine(1.0, 0.999) as ref, bh.PhysicsEngine(1.0, 0.999) as eng:
        want = {}
        for c in set(cams):
            want[c] = _bits(_frame(bh, ref, c, arith=arith, schedule=bh.SCHEDULE_SLOT_ORDER))
        assert len({w[2].sum() for w in want.values()}) == len(want)
        side = torch.cuda.Stream()
        held = []
        for rep, c in enumerate(cams):       # queued back to back: two frames in flight on two streams
            st = side if rep % 2 else torch.cuda.current_stream()
            with torch.cuda.stream(st):
                held.append((c, _frame(bh, eng, c, stream=st.cuda_stream, arith=arith)))
        for rep, (c, out) in enumerate(held):
            got = _bits(out)
            for k, (g, w) in enumerate(zip(got, want[c])):
                assert np.array_equal(g, w), (rep, k)
        st = eng.frame_stats()
        assert st.accepted_steps == int(want[cams[-1]][2].sum())


def test_small_whole_frames_take_a_measured_order_too(engine_mod):
    """Whole frames from 65 536 rays on run one-wave blocks longest-first (engine_types.hpp kSegOrderMinRays; before the
    resolution sweep of round 6 only frames of 1.5 M rays and more did): 640x360 and 1280x720 frames with moving cameras on
    two streams against slot-order frames, bit for bit; a 192x108 frame (below the bound) the same."""
    import torch
    bh = engine_mod
    cams = [_eye(20.0, 97.0), _eye(12.0, 80.0), _eye(20.0, 97.0), _eye(6.0, 91.0), _eye(20.0, 97.0)]
    for w, h in ((640, 360), (1280, 720), (192, 108)):
        n = w * h
        with bh.PhysicsEngine(1.0, 0.999) as ref, bh.PhysicsEngine(1.0, 0.999) as eng:
            def frame(e, c, st=None, **kw):
                out = dict(rgba=torch.full((n, 4), -1.0, dtype=torch.float32, device="cuda"),
                           final_state=torch.zeros(n, 8, dtype=torch.float64, device="cuda"),
                           steps=torch.zeros(n, dtype=torch.int32, device="cuda"))
                p = bh.render_params(w, h, arith=bh.ARITH_FAST, tolerance=1e-7, max_steps=600, **kw)
                e.render_frame_device(bh.camera_look_at(c, aspect=w / h), p, stream=st, **out)
                return out
            want = {}
            for c in set(cams):
                o = frame(ref, c, schedule=bh.SCHEDULE_SLOT_ORDER)
                torch.cuda.synchronize()
                want[c] = [o[k].cpu().numpy() for k in ("rgba", "final_state", "steps")]
            side = torch.cuda.Stream()
            held = []
            for rep, c in enumerate(cams):
                st = side if rep % 2 else torch.cuda.current_stream()
                with torch.cuda.stream(st):
                    held.append((c, frame(eng, c, st=st.cuda_stream)))
            torch.cuda.synchronize()
            for rep, (c, o) in enumerate(held):
                for k, wv in zip(("rgba", "final_state", "steps"), want[c]):
                    g = o[k].cpu().numpy()
                    assert np.array_equal(g.view(np.uint8), wv.view(np.uint8)), (w, h, rep, k)


def test_compacting_schedule_equals_the_one_launch_frame(engine_mod):
    bh = engine_mod
    c = _eye(10.0, 97.0)
    with bh.PhysicsEngine(1.0, 0.999) as eng:
        want = _bits(_frame(bh, eng, c, schedule=bh.SCHEDULE_SLOT_ORDER))
        for k in (16, 7, 64, 1000):
            for rep in range(2):
                got = _bits(_frame(bh, eng, c, segment_tries=k))
                st = eng.frame_stats()
                for j, (g, w) in enumerate(zip(got, want)):
                    assert np.array_equal(g, w), (k, rep, j)
                assert st.accepted_steps == int(want[2].sum()) and st.launches >= 1
        with pytest.raises(bh.GravitasError, match="schedule"):
            _frame(bh, eng, c, schedule=7)


def test_compacting_schedule_with_a_stale_head_start_forecast(engine_mod):
    """The compacting schedule gives the top eighth of the PREVIOUS frame's waves (longest first) a head start in one
    launch beside the chain (engine.hip run_segments).  The forecast is stale by construction when the camera moves --
    rays in the head that are short, long rays left in the chain --: the frame must not care."""
    import torch
    bh = engine_mod
    cams = [_eye(20.0, 97.0), _eye(7.0, 80.0), _eye(20.0, 97.0), _eye(40.0, 30.0), _eye(7.0, 80.0), _eye(20.0, 97.0)]
    with bh.PhysicsEngine(1.0, 0.999) as ref, bh.PhysicsEngine(1.0, 0.999) as eng:
        want = {c: _bits(_frame(bh, ref, c, schedule=bh.SCHEDULE_SLOT_ORDER)) for c in set(cams)}
        side = torch.cuda.Stream()
        for k in (16, 48):
            held = []
            for rep, c in enumerate(cams):      # queued back to back on two streams: two frames in flight
                st = side if rep % 2 else torch.cuda.current_stream()
                with torch.cuda.stream(st):
                    held.append((c, _frame(bh, eng, c, stream=st.cuda_stream, segment_tries=k)))
            for rep, (c, out) in enumerate(held):
                got = _bits(out)
                for j, (g, w) in enumerate(zip(got, want[c])):
                    assert np.array_equal(g, w), (k, rep, j)
