"""The multi-GPU frame behind the C ABI (grv_engine_create_multi*, csrc/engine_multi.hip) on ONE GPU:
G virtual ranks share cuda:0, each with its own engine, host thread and two streams; every rank
renders its round-robin tile share, the peer-copy transport pushes the shares into rank 0's receive
slots, rank 0 de-interleaves.  The assembled image must equal grv_render_frame_device's whole frame
bit for bit -- same kernels, same rays, another schedule.  The RCCL transport is walked with one
device (communicator set-up, one self send/recv group per frame); more than one rank per device is
something RCCL refuses, and the pool hands out one-GPU boxes."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EYE = (60.0 * np.sin(np.deg2rad(97.0)), 60.0 * np.cos(np.deg2rad(97.0)), 0.0)


@pytest.fixture(scope="module")
def bh(engine_mod):
    return engine_mod


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _whole(bh, torch, w, h, spin=0.999, **kw):
    cam = bh.camera_look_at(EYE, aspect=w / h)
    p = bh.render_params(w, h, **kw)
    with bh.PhysicsEngine(1.0, spin) as e:
        out = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0")
        e.render_frame_device(cam, p, rgba=out)
        st = e.frame_stats()
        torch.cuda.synchronize()
    return cam, p, out, st


@pytest.mark.parametrize("ranks,w,h,arith", [(2, 200, 130, 0), (4, 512, 288, 1), (8, 960, 540, 1), (3, 333, 77, 1),
                                             (8, 64, 64, 1)])
def test_virtual_ranks_assemble_the_whole_frame_bitwise(bh, torch, ranks, w, h, arith):
    cam, p, want, wst = _whole(bh, torch, w, h, arith=arith)
    with bh.MultiEngine(1.0, 0.999, virtual_ranks=ranks) as m:
        assert m.ranks == ranks and m.rank_devices() == [0] * ranks and m.transport == bh.TRANSPORT_PEER_COPY
        got = torch.full((h, w, 4), -7.0, dtype=torch.float32, device="cuda:0")
        m.render_frame_device(cam, p, got)
        st = m.frame_stats()
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))
        assert (st.rays, st.accepted_steps, st.rkf_tries) == (wst.rays, wst.accepted_steps, wst.rkf_tries)
        assert list(st.term_count) == list(wst.term_count) and st.max_drift == wst.max_drift
        # host-pointer entry
        img, st2 = m.render_frame(cam, p)
        assert np.array_equal(img.view(np.uint32), want.cpu().numpy().view(np.uint32))
        assert st2.accepted_steps == wst.accepted_steps


def test_frames_in_flight_and_changing_cameras(bh, torch):
    """Eight frames queued back to back (even / odd frames on the two stream sets, no host wait
    between them), each with its own camera and its own output buffer: every image equals the
    single-engine render of that camera; accumulated counters are the sums."""
    w, h, ranks = 384, 216, 4
    eyes = [(60.0 * np.sin(t), 60.0 * np.cos(t), 3.0 * k) for k, t in enumerate(np.deg2rad(np.linspace(60, 120, 8)))]
    p = bh.render_params(w, h, arith=1)
    want, steps = [], 0
    with bh.PhysicsEngine(1.0, 0.999) as e:
        for eye in eyes:
            o = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0")
            e.render_frame_device(bh.camera_look_at(eye, aspect=w / h), p, rgba=o)
            steps += e.frame_stats().accepted_steps
            want.append(o)
    with bh.MultiEngine(1.0, 0.999, virtual_ranks=ranks) as m:
        m.stats_accumulate(True)
        m.frame_stats_reset()
        got = [torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0") for _ in eyes]
        for eye, o in zip(eyes, got):
            m.render_frame_device(bh.camera_look_at(eye, aspect=w / h), p, o)
        st = m.frame_stats()
        torch.cuda.synchronize()
        for k in range(len(eyes)):
            assert torch.equal(got[k].view(torch.int32), want[k].view(torch.int32)), k
        assert st.accepted_steps == steps
        # the frame size may change between frames (buffers grow on demand)
        cam2, p2, want2, _ = _whole(bh, torch, 700, 400, arith=1)
        big = torch.zeros(400, 700, 4, dtype=torch.float32, device="cuda:0")
        m.render_frame_device(cam2, p2, big)
        m.synchronize()
        assert torch.equal(big.view(torch.int32), want2.view(torch.int32))


def test_update_params_reaches_every_rank(bh, torch):
    w, h = 256, 144
    cam, p, want, _ = _whole(bh, torch, w, h, spin=0.5, arith=1, disk_profile=1)
    with bh.MultiEngine(1.0, 0.999, virtual_ranks=3) as m:
        got = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0")
        m.render_frame_device(cam, p, got)
        m.update_params(1.0, 0.5)
        m.render_frame_device(cam, p, got)
        m.synchronize()
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))


def test_wgsl_march_over_virtual_ranks(bh, torch):
    """BASELINE configs[3]'s kernel (f32 compute march, packed two-rays-per-lane form) over ranks."""
    w, h = 512, 288
    cam = bh.camera_look_at(EYE, aspect=w / h)
    for arith in (bh.ARITH_STRICT, bh.ARITH_FAST_PACKED):
        wp = bh.wgsl_params(w, h, cam, 1.0, 0.999, max_steps=300, arith=arith)
        with bh.PhysicsEngine(1.0, 0.999) as e:
            want = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0")
            total = e.render_frame_wgsl(wp, want)
        with bh.MultiEngine(1.0, 0.999, virtual_ranks=4) as m:
            got = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0")
            m.render_frame_wgsl_device(wp, got)
            st = m.frame_stats()
            assert torch.equal(got.view(torch.int32), want.view(torch.int32))
            assert st.accepted_steps == total


def _half_rounded(torch, img):
    """channel by channel through binary16, round to nearest even: what an rgba16float target stores"""
    return img.to(torch.float16).to(torch.float32)


@pytest.mark.parametrize("ranks,w,h", [(1, 200, 130), (2, 200, 130), (4, 512, 288), (8, 960, 540), (3, 333, 77)])
def test_rgba16f_exchange_assembles_the_half_rounded_frame_bitwise(bh, torch, ranks, w, h):
    """grv_multi_set_exchange_format(RGBA16F): every rank narrows its share to the compute pass's own
    output format (rgba16float storage texture, src/rendering/webgpu/renderer.ts:163-176) before the ONE
    exchange, rank 0 widens while it de-interleaves.  The image is the one-device frame rounded through
    binary16 -- bit for bit, for any rank count --, the f32 frame and the f32 compute march alike, with
    frames in flight; the bytes that cross devices are half of the RGBA32F exchange's."""
    cam, p, want, wst = _whole(bh, torch, w, h, arith=1)
    want16 = _half_rounded(torch, want)
    assert not torch.equal(want16, want)   # the rounding is visible: the test can tell the formats apart
    wp = bh.wgsl_params(w, h, cam, 1.0, 0.999, max_steps=200, arith=bh.ARITH_FAST_PACKED)
    with bh.PhysicsEngine(1.0, 0.999) as e:
        wantw = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0")
        e.render_frame_wgsl(wp, wantw)
    make = (lambda: bh.MultiEngine(1.0, 0.999, devices=[0])) if ranks == 1 else \
        (lambda: bh.MultiEngine(1.0, 0.999, virtual_ranks=ranks))
    with make() as m:
        assert m.exchange_format == bh.EXCHANGE_RGBA32F
        full = m.exchange_bytes_per_frame(w, h)
        m.set_exchange_format(bh.EXCHANGE_RGBA16F)
        assert m.exchange_format == bh.EXCHANGE_RGBA16F and m.exchange_bytes_per_frame(w, h) * 2 == full
        assert (full > 0) == (ranks > 1)
        outs = [torch.full((h, w, 4), -7.0, dtype=torch.float32, device="cuda:0") for _ in range(4)]
        for o in outs:                      # four frames queued back to back: both buffer parities, twice
            m.render_frame_device(cam, p, o)
        st = m.frame_stats()
        for o in outs:
            assert torch.equal(o.view(torch.int32), want16.view(torch.int32))
        assert st.accepted_steps == wst.accepted_steps
        got = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0")
        m.render_frame_wgsl_device(wp, got)
        m.synchronize()
        assert torch.equal(got.view(torch.int32), _half_rounded(torch, wantw).view(torch.int32))
        # host-pointer entry, then back to RGBA32F on the same handle (buffers are re-laid out)
        img, _ = m.render_frame(cam, p)
        assert np.array_equal(img.view(np.uint32), want16.cpu().numpy().view(np.uint32))
        m.set_exchange_format(bh.EXCHANGE_RGBA32F)
        m.render_frame_device(cam, p, got)
        m.synchronize()
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))
        with pytest.raises(bh.GravitasError):
            m.set_exchange_format(7)


def test_single_rank_handle_is_the_plain_frame(bh, torch):
    cam, p, want, wst = _whole(bh, torch, 320, 180, arith=1)
    with bh.MultiEngine(1.0, 0.999, devices=[0]) as m:
        assert m.ranks == 1
        got = torch.zeros(180, 320, 4, dtype=torch.float32, device="cuda:0")
        m.render_frame_device(cam, p, got)
        assert m.frame_stats().accepted_steps == wst.accepted_steps
        assert torch.equal(got.view(torch.int32), want.view(torch.int32))


def test_bad_requests_are_refused(bh, torch):
    L = bh.load_library()
    h = C.c_void_p()
    assert L.grv_engine_create_multi(1.0, 0.5, 0, 0, C.byref(h)) == 1           # empty mask
    assert L.grv_engine_create_multi(1.0, 0.5, 1 << 40, 0, C.byref(h)) == 2     # no such device
    assert L.grv_engine_create_multi_virtual(1.0, 0.5, 0, 0, C.byref(h)) == 1   # zero ranks
    with bh.MultiEngine(1.0, 0.999, virtual_ranks=2) as m:
        cam = bh.camera_look_at(EYE, aspect=2.0)
        p = bh.render_params(128, 64, arith=1, tile_world=2, tile_rank=1)
        out = torch.zeros(64, 128, 4, dtype=torch.float32, device="cuda:0")
        with pytest.raises(bh.GravitasError, match="tile_world"):
            m.render_frame_device(cam, p, out)
        with pytest.raises(bh.GravitasError):
            m.render_frame_device(cam, bh.render_params(128, 64, arith=7), out)
        # the handle is still usable afterwards
        m.render_frame_device(cam, bh.render_params(128, 64, arith=1), out)
        m.synchronize()


_RCCL_WALK = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
import blackhole_simulation_amd as bh
EYE = (60.0 * np.sin(np.deg2rad(97.0)), 60.0 * np.cos(np.deg2rad(97.0)), 0.0)
w, h = 320, 180
cam = bh.camera_look_at(EYE, aspect=w / h)
p = bh.render_params(w, h, arith=1)
with bh.PhysicsEngine(1.0, 0.999) as e:
    want = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0")
    e.render_frame_device(cam, p, rgba=want)
    torch.cuda.synchronize()
with bh.MultiEngine(1.0, 0.999, devices=[0], transport=bh.TRANSPORT_RCCL) as m:
    assert m.transport == bh.TRANSPORT_RCCL
    try:
        m.test_self_exchange(True)
        raise SystemExit("the hook answered while locked")
    except bh.GravitasError as e:
        assert "locked" in str(e)
    bh.unlock_test_hooks()
    m.test_self_exchange(True)
    for k in range(3):
        got = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0")
        m.render_frame_device(cam, p, got)
        m.synchronize()
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), k
    # the same walk with binary16 on the wire (ncclHalf elements)
    m.set_exchange_format(bh.EXCHANGE_RGBA16F)
    want16 = want.to(torch.float16).to(torch.float32)
    for k in range(3):
        got = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0")
        m.render_frame_device(cam, p, got)
        m.synchronize()
        assert torch.equal(got.view(torch.int32), want16.view(torch.int32)), ("rgba16f", k)
    # an ncclSend that fails inside the group: the call reports it, the group is closed, the next frame is whole
    m.set_exchange_format(bh.EXCHANGE_RGBA32F)
    m.test_inject_fault(bh.FAULT_SEND, 0)
    got = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0")
    try:
        m.render_frame_device(cam, p, got)
        raise SystemExit("the injected ncclSend failure went unnoticed")
    except bh.GravitasError as e:
        assert "injected ncclSend failure" in str(e) and "group closed" in str(e), str(e)
    for k in range(2):
        got = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0")
        m.render_frame_device(cam, p, got)
        m.synchronize()
        assert torch.equal(got.view(torch.int32), want.view(torch.int32)), ("after the fault", k)
print("RCCL_WALK_OK")
"""


def test_rccl_transport_walk_on_one_device():
    """ncclCommInitAll on one device and, with grv_multi_test_self_exchange, rank 0's share travelling
    through one ncclSend / ncclRecv group per frame before the unpack: every RCCL call of the
    transport runs, on the one device this pool has."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_WALK % {"root": ROOT}], capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_WALK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def _bench_native(*args, flag=("--native",)):
    import json
    env = dict(os.environ, GRV_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flag) + list(args),
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_native_drives_the_c_abi_multi_handle():
    """bench.py --native: one process, grv_engine_create_multi (here: virtual ranks on the one GPU),
    same frame and the same accepted steps per frame as the one-rank run."""
    one = _bench_native("--gpus", "1", "--width", "640", "--height", "360", "--steps", "4", "--warmup", "1")
    # `--launcher native` is the explicit spelling (a bare `python bench.py --gpus 4` starts four
    # torch.distributed ranks until the handle has run on two real devices)
    four = _bench_native("--gpus", "4", "--width", "640", "--height", "360", "--steps", "4", "--warmup", "1",
                         flag=("--launcher", "native"))
    assert one["n_gpus"] == 1 and four["n_gpus"] == 4 and four["ranks"] == 4 and four["rank_devices"] == [0] * 4
    assert "native" in four["launcher"] and four["transport"] == "peer_copy" and four["rccl_version"] is None
    rk = four["rank_integrate_ms"]
    assert len(rk["per_rank"]) == 4 and 0 < rk["min"] <= rk["max"]
    assert four["config"]["accepted_steps_per_frame"] == one["config"]["accepted_steps_per_frame"]
    assert four["config"]["virtual_ranks_on_one_device"] and four["config"]["frames_in_flight"] == 2
    assert four["roofline"]["avg_launch_ms"] > 0 and four["value"] > 0
    c4 = _bench_native("--gpus", "2", "--config", "c4", "--width", "512", "--height", "288", "--steps", "3",
                       "--warmup", "1")
    assert c4["dtype"] == "f32" and c4["n_gpus"] == 2 and c4["config"]["accepted_steps_per_frame"] > 0
    # the f32 march is timed rank by rank too (grv_engine_profile_shader_frames)
    rk = c4["rank_integrate_ms"]
    assert len(rk["per_rank"]) == 2 and 0 < rk["min"] <= rk["max"]
    for ln in (one, four, c4):
        rf = ln["roofline"]
        assert rf["bound"] in ("fp64_valu", "fp32_valu") and rf["unit"] == "TFLOP/s"
        assert isinstance(rf["frac"], float) and 0 < rf["frac"] <= 1.0, rf
        assert rf["flops_source"] in ("counted", "algorithmic") and "hbm_nominal" in rf


def test_injected_faults_are_reported_and_the_next_frame_is_whole(bh):
    """grv_multi_test_inject_fault: a rank whose render call fails on frame k, and a rank whose
    push to rank 0 fails -- the frame call returns an error that names the rank and the cause, other ranks'
    queued work is harmless, and the SAME handle renders the following frames bit-equal to one device, with frames
    in flight on two streams.  (The ncclSend fault runs in test_rccl_transport_walk_on_one_device: RCCL needs a
    process of its own on this pool.)"""
    import torch
    bh.unlock_test_hooks()
    w, h = 640, 360
    cam = bh.camera_look_at(EYE, aspect=w / h)
    p = bh.render_params(w, h, arith=bh.ARITH_FAST)
    with bh.PhysicsEngine(1.0, 0.999) as e:
        want = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0")
        e.render_frame_device(cam, p, rgba=want)
        torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for G in (2, 4, 8):
        with bh.MultiEngine(1.0, 0.999, virtual_ranks=G) as m:
            outs = [torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0") for _ in range(2)]
            frame = 0
            for kind, rank, text in ((bh.FAULT_RENDER, G - 1, "injected render failure"),
                                     (bh.FAULT_PEER_COPY, 1, "injected peer-copy failure"),
                                     (bh.FAULT_RENDER, 0, "injected render failure")):
                for _ in range(3):  # healthy frames in flight first
                    with torch.cuda.stream(streams[frame % 2]):
                        m.render_frame_device(cam, p, outs[frame % 2], stream=streams[frame % 2].cuda_stream)
                    frame += 1
                m.test_inject_fault(kind, rank)
                with pytest.raises(bh.GravitasError) as ei:
                    with torch.cuda.stream(streams[frame % 2]):
                        m.render_frame_device(cam, p, outs[frame % 2], stream=streams[frame % 2].cuda_stream)
                assert text in str(ei.value) and ("rank %d" % rank) in str(ei.value), str(ei.value)
                for _ in range(3):  # ... and whole frames after it, both parities
                    with torch.cuda.stream(streams[frame % 2]):   # the fill on the frame's own stream: the null stream
                        outs[frame % 2].fill_(-1.0)               # is not ordered with a non-blocking one
                        m.render_frame_device(cam, p, outs[frame % 2], stream=streams[frame % 2].cuda_stream)
                    m.synchronize()
                    torch.cuda.synchronize()
                    got = outs[frame % 2]
                    assert torch.equal(got.view(torch.int32), want.view(torch.int32)), \
                        (G, kind, frame, int((got != want).any(dim=2).sum()), int((got == -1.0).all(dim=2).sum()))
                    frame += 1
            with pytest.raises(bh.GravitasError, match="out of range"):
                m.test_inject_fault(bh.FAULT_RENDER, G)
