"""The image plane over SEVERAL REAL devices (tile grid of
physics-engine/_legacy_src/tiling.rs:38-56 dealt round-robin): these tests switch themselves on
wherever torch sees two or more HIP devices and are skipped, one by one, on a smaller box -- the
pool this repo is developed on hands out one-GPU boxes, so the first multi-GPU node that runs the
suite is also the first to run them.  Every assembled image must equal the one-device render of the
same camera BIT FOR BIT (same kernels, same rays, another schedule and a real exchange):
  * the C ABI's multi-GPU handle (grv_engine_create_multi) with the RCCL send/recv group and with
    peer copies, G = 2, 4, 8, f64 frame (both contracts) and the f32 compute march of configs[3],
    six frames in flight with a moving camera;
  * torch.distributed ranks (one process per GPU, nccl backend = RCCL) through
    blackhole_simulation_amd.distributed;
  * renderFrame({devices: G}) from Node through the N-API addon;
  * bench.py --gpus G under both launchers: G ranks on G distinct devices, the one-GPU workload.
"""
import ctypes as C
import json
import os
import shutil
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EYE = (60.0 * np.sin(np.deg2rad(97.0)), 60.0 * np.cos(np.deg2rad(97.0)), 0.0)


def _device_count():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


N_DEV = _device_count()


def needs(g):
    return pytest.mark.skipif(N_DEV < g, reason="needs %d HIP devices, %d visible" % (g, N_DEV))


GS = [pytest.param(g, marks=needs(g)) for g in (2, 4, 8)]


@pytest.fixture(scope="module")
def bh(engine_mod):
    return engine_mod


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _whole(bh, torch, w, h, eye=EYE, spin=0.999, **kw):
    cam = bh.camera_look_at(eye, aspect=w / h)
    p = bh.render_params(w, h, **kw)
    with bh.PhysicsEngine(1.0, spin) as e:
        out = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0")
        e.render_frame_device(cam, p, rgba=out)
        st = e.frame_stats()
        torch.cuda.synchronize()
    return cam, p, out, st


def _transports(bh):
    return [("rccl", bh.TRANSPORT_RCCL), ("peer_copy", bh.TRANSPORT_PEER_COPY)]


@pytest.mark.parametrize("G", GS)
@pytest.mark.parametrize("w,h,arith", [(960, 540, 1), (333, 211, 0)])
def test_real_devices_assemble_the_whole_f64_frame_bitwise(bh, torch, G, w, h, arith):
    cam, p, want, wst = _whole(bh, torch, w, h, arith=arith)
    images = {}
    for name, tr in _transports(bh):
        with bh.MultiEngine(1.0, 0.999, devices=list(range(G)), transport=tr) as m:
            assert m.ranks == G and m.rank_devices() == list(range(G)) and m.transport == tr
            got = torch.full((h, w, 4), -7.0, dtype=torch.float32, device="cuda:0")
            m.render_frame_device(cam, p, got)
            st = m.frame_stats()
            assert torch.equal(got.view(torch.int32), want.view(torch.int32)), name
            assert (st.rays, st.accepted_steps, st.rkf_tries) == (wst.rays, wst.accepted_steps, wst.rkf_tries)
            assert list(st.term_count) == list(wst.term_count) and st.max_drift == wst.max_drift
            # every rank did part of the work, on its own device
            shares = [m.rank_frame_stats(r).rays for r in range(G)]
            assert sum(shares) == wst.rays and (min(shares) > 0 or w * h < 4096 * G)
            img, st2 = m.render_frame(cam, p)  # host-pointer entry
            assert np.array_equal(img.view(np.uint32), want.cpu().numpy().view(np.uint32))
            assert st2.accepted_steps == wst.accepted_steps
            images[name] = got.clone()
    assert torch.equal(images["rccl"].view(torch.int32), images["peer_copy"].view(torch.int32))


@pytest.mark.parametrize("G", GS)
def test_rgba16f_exchange_between_real_devices(bh, torch, G):
    """The exchange in the compute pass's own rgba16float format (renderer.ts:163-176) over RCCL
    (ncclHalf) and peer copies: the half-rounded one-device frame bit for bit, half the bytes."""
    w, h = 960, 540
    cam, p, want, wst = _whole(bh, torch, w, h, arith=1)
    want16 = want.to(torch.float16).to(torch.float32)
    for name, tr in _transports(bh):
        with bh.MultiEngine(1.0, 0.999, devices=list(range(G)), transport=tr) as m:
            full = m.exchange_bytes_per_frame(w, h)
            m.set_exchange_format(bh.EXCHANGE_RGBA16F)
            assert m.exchange_bytes_per_frame(w, h) * 2 == full and full > 0
            outs = [torch.full((h, w, 4), -7.0, dtype=torch.float32, device="cuda:0") for _ in range(4)]
            for o in outs:
                m.render_frame_device(cam, p, o)
            m.synchronize()
            for o in outs:
                assert torch.equal(o.view(torch.int32), want16.view(torch.int32)), name
            assert m.frame_stats().accepted_steps >= wst.accepted_steps


@pytest.mark.parametrize("G", GS)
def test_auto_transport_between_real_devices_is_rccl(bh, torch, G):
    with bh.MultiEngine(1.0, 0.999, devices=list(range(G))) as m:
        assert m.transport == bh.TRANSPORT_RCCL
    v, why = bh.rccl_probe()
    assert v >= 20000, why


@pytest.mark.parametrize("G", GS)
def test_six_frames_in_flight_with_a_moving_camera(bh, torch, G):
    """Frames queued back to back (even / odd frames alternate the two stream and buffer sets, no
    host wait in between), each with its own camera and output buffer, both transports."""
    w, h = 640, 360
    eyes = [(60.0 * np.sin(t), 60.0 * np.cos(t), 3.0 * k) for k, t in enumerate(np.deg2rad(np.linspace(60, 120, 6)))]
    p = bh.render_params(w, h, arith=1)
    want, steps = [], 0
    with bh.PhysicsEngine(1.0, 0.999) as e:
        for eye in eyes:
            o = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0")
            e.render_frame_device(bh.camera_look_at(eye, aspect=w / h), p, rgba=o)
            steps += e.frame_stats().accepted_steps
            want.append(o)
    for name, tr in _transports(bh):
        with bh.MultiEngine(1.0, 0.999, devices=list(range(G)), transport=tr) as m:
            m.stats_accumulate(True)
            m.frame_stats_reset()
            got = [torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0") for _ in eyes]
            for eye, o in zip(eyes, got):
                m.render_frame_device(bh.camera_look_at(eye, aspect=w / h), p, o)
            st = m.frame_stats()
            torch.cuda.synchronize()
            for k in range(len(eyes)):
                assert torch.equal(got[k].view(torch.int32), want[k].view(torch.int32)), (name, k)
            assert st.accepted_steps == steps
            # the frame size may change between frames (buffers grow on demand), and update_params
            # reaches every rank
            m.update_params(1.0, 0.5)
            cam2, p2, want2, _ = _whole(bh, torch, 700, 400, spin=0.5, arith=1)
            big = torch.zeros(400, 700, 4, dtype=torch.float32, device="cuda:0")
            m.render_frame_device(cam2, p2, big)
            m.synchronize()
            assert torch.equal(big.view(torch.int32), want2.view(torch.int32)), name


@pytest.mark.parametrize("G", GS)
def test_config4_march_over_real_devices(bh, torch, G):
    """BASELINE configs[3]: the f32 compute march (shader order and the packed two-rays-per-lane
    form) tiled over G devices, RCCL and peer copies."""
    w, h = 1024, 576
    cam = bh.camera_look_at(EYE, aspect=w / h)
    for arith in (bh.ARITH_STRICT, bh.ARITH_FAST_PACKED):
        wp = bh.wgsl_params(w, h, cam, 1.0, 0.999, max_steps=300, arith=arith)
        with bh.PhysicsEngine(1.0, 0.999) as e:
            want = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0")
            total = e.render_frame_wgsl(wp, want)
        for name, tr in _transports(bh):
            with bh.MultiEngine(1.0, 0.999, devices=list(range(G)), transport=tr) as m:
                got = torch.zeros(h, w, 4, dtype=torch.float32, device="cuda:0")
                for _ in range(3):  # both buffer parities
                    got.zero_()
                    m.render_frame_wgsl_device(wp, got)
                    m.synchronize()
                    assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (arith, name)
                assert m.frame_stats().accepted_steps == total


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _dist_worker(rank, world, port, w, h, arith, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import blackhole_simulation_amd as bh
    from blackhole_simulation_amd import distributed as D

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        cam = bh.camera_look_at(EYE, aspect=w / h)
        params = bh.render_params(w, h, arith=arith)
        with bh.PhysicsEngine(1.0, 0.999, device=rank) as eng:
            images = []
            for _ in range(2):
                img, st = D.render_frame_distributed(eng, cam, params,
                                                     stream=torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                if rank == 0:
                    images.append(img.cpu().numpy().copy())
            steps = torch.tensor([float(st.accepted_steps)], dtype=torch.float64, device="cuda")
            dist.all_reduce(steps)
            if rank == 0:
                whole = torch.zeros(w * h, 4, dtype=torch.float32, device="cuda")
                eng.render_frame_device(cam, params, rgba=whole, stream=torch.cuda.current_stream().cuda_stream)
                wst = eng.frame_stats()
                torch.cuda.synchronize()
                np.savez(out_path, whole=whole.cpu().numpy().reshape(h, w, 4), images=np.stack(images),
                         steps_sum=float(steps.item()), steps_whole=float(wst.accepted_steps))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("G", GS)
def test_torch_distributed_ranks_gather_over_rccl(tmp_path, engine_mod, G):
    """One process per GPU, nccl (= RCCL) backend: render_frame_distributed's gather of finished
    tiles to rank 0 and the de-interleave give the whole frame, twice in a row."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "r.npz")
    mp.spawn(_dist_worker, args=(G, _free_port(), 800, 450, 1, out), nprocs=G, join=True)
    r = np.load(out)
    for img in r["images"]:
        assert np.array_equal(img.view(np.uint32), r["whole"].view(np.uint32))
    assert r["steps_sum"] == r["steps_whole"]


@pytest.mark.parametrize("G", GS)
def test_render_frame_from_node_over_real_devices(G):
    node = shutil.which("node")
    addon = os.path.join(ROOT, "napi", "blackhole_physics.node")
    if node is None or not os.path.exists(addon):
        pytest.skip("node or the built addon is not available")
    r = subprocess.run([node, os.path.join(ROOT, "napi", "multi.js"), str(G)], capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-3000:] + r.stdout[-1000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["G"] == G and len(res["frames"]) == 6 and res["async_equal"]
    for f in res["frames"]:
        assert f["equal"] and f["steps_equal"] and f["devices"] == G and f["transport"] == "rccl", f


def _bench(G, *args, launcher):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE",
                                                            "GRV_BENCH_ONE_DEVICE", "GRV_BENCH_BACKEND")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if launcher == "driver":  # exactly as the round-end driver starts N > 1
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(G),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")]
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")]
        if launcher != "bare":
            cmd += ["--launcher", launcher]
    cmd += ["--gpus", str(G)] + list(args)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("G", GS)
def test_bench_runs_G_ranks_on_G_devices_under_both_launchers(G):
    """`python bench.py --gpus G` bare (starts its own G torch.distributed ranks), with --launcher
    native (the C ABI's multi-GPU handle, one process), and under the driver's own torch.distributed.run
    command: G ranks, G distinct devices, RCCL, the accepted steps of the one-GPU frame, per-rank
    integrate times in the line."""
    size = ["--width", "1280", "--height", "720", "--steps", "4", "--warmup", "1"]
    one = _bench(1, *size, "--no-cpu-baseline", launcher="bare")
    for launcher in ("bare", "native", "driver"):
        ln = _bench(G, *size, launcher=launcher)
        assert ln["n_gpus"] == G and ln["ranks"] == G, launcher
        assert sorted(ln["rank_devices"]) == list(range(G)) and len(set(ln["rank_devices"])) == G, launcher
        assert ln["transport"] == "rccl" and ln["rccl_version"], launcher
        assert ln["config"]["accepted_steps_per_frame"] == one["config"]["accepted_steps_per_frame"], launcher
        assert ln["scaling"] == "strong" and "split over %d GPUs" % G in ln["config"]["workload"]
        rk = ln["rank_integrate_ms"]
        assert len(rk["per_rank"]) == G and 0 < rk["min"] <= rk["max"], launcher
        assert ("native" in ln["launcher"]) == (launcher == "native")
    c4 = _bench(G, "--config", "c4", "--width", "1024", "--height", "576", "--steps", "3", "--warmup", "1",
                launcher="native")
    c4_one = _bench(1, "--config", "c4", "--width", "1024", "--height", "576", "--steps", "3", "--warmup", "1",
                    "--no-cpu-baseline", launcher="bare")
    assert c4["n_gpus"] == G and c4["dtype"] == "f32"
    assert len(c4["rank_integrate_ms"]["per_rank"]) == G and c4["rank_integrate_ms"]["min"] > 0
    assert c4["config"]["accepted_steps_per_frame"] == c4_one["config"]["accepted_steps_per_frame"]


@needs(2)
def test_rccl_failure_is_loud_not_a_silent_peer_copy(bh, torch):
    """A handle that asks for RCCL and cannot have it is refused with the loader's message; the
    bench exits non-zero without a JSON line.  (Own process: the library binds librccl once.)"""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import blackhole_simulation_amd as bh\n"
            "try:\n"
            "    bh.MultiEngine(1.0, 0.9, devices=[0, 1])\n"
            "    print('OPENED')\n"
            "except bh.GravitasError as e:\n"
            "    print('REFUSED', e)\n" % ROOT)
    env = dict(os.environ, GRV_RCCL_LIBRARY="/nonexistent/librccl.so.1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0 and "REFUSED" in r.stdout and "dlopen(/nonexistent/librccl.so.1)" in r.stdout, r.stdout + r.stderr
    env = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--width", "256", "--height", "144"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 3 and "RCCL transport unavailable" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_the_gate_itself():
    """On a one-GPU box everything above is skipped and says why; this test records the device count
    in the report so a reader can tell which of the two happened."""
    print("HIP devices visible: %d" % N_DEV)
    assert N_DEV >= 1
    if N_DEV == 1:
        h = C.c_void_p()
        import blackhole_simulation_amd as bh
        assert bh.load_library().grv_engine_create_multi(1.0, 0.5, 0b11, bh.TRANSPORT_RCCL, C.byref(h)) == 2
        assert b"device 1 asked for, 1 visible" in bh.load_library().grv_multi_create_error()
