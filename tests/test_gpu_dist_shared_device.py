"""The N > 1 path on ONE GPU: several processes share cuda:0, render their tile shares with the HIP
engine into the gather's send buffers and exchange the device buffers through gloo (RCCL refuses
two ranks on one device; the pool hands out one-GPU boxes).  Everything but the transport is the
path bench.py and render_frame_distributed run on eight GPUs: rank_params, the packed render
targets, TileGather (synchronous and pipelined, two frames in flight on two streams), the
de-interleave kernel on rank 0, the timing reductions.  The assembled image must equal a
whole-frame render bit for bit."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, w, h, arith, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import blackhole_simulation_amd as bh
    from blackhole_simulation_amd import distributed as D

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        th = np.deg2rad(97.0)
        eye = (60.0 * np.sin(th), 60.0 * np.cos(th), 0.0)
        cam = bh.camera_look_at(eye, aspect=w / h)
        params = bh.render_params(w, h, arith=arith)
        rp = D.rank_params(params, world, rank)
        with bh.PhysicsEngine(1.0, 0.999, device=0) as eng:
            n_local = eng.frame_ray_count(rp)

            def dev_unpack(rparams, r, packed, image):
                eng.unpack_tiles_device(rparams, r, packed, image, 16, torch.cuda.current_stream().cuda_stream)

            # synchronous form
            tg = D.TileGather(params, world, rank, 4, torch.float32, dev)
            eng.render_frame_device(cam, rp, rgba=tg.local_view(n_local),
                                    stream=torch.cuda.current_stream().cuda_stream)
            img = tg.run(dev_unpack)
            torch.cuda.synchronize()
            images = [img.cpu().numpy().copy()] if rank == 0 else []
            # pipelined form, even / odd frames on two streams (bench.py's default for N > 1):
            # four frames of the same camera must come back identical to the synchronous one
            tg2 = D.TileGather(params, world, rank, 4, torch.float32, dev).enable_pipeline()
            streams = [torch.cuda.Stream(), torch.cuda.Stream()]
            eng.stats_accumulate(True)
            eng.frame_stats_reset(streams[0].cuda_stream)
            torch.cuda.synchronize()
            for f in range(4):
                with torch.cuda.stream(streams[f % 2]):
                    target = tg2.pipelined_view(f, n_local)
                    eng.render_frame_device(cam, rp, rgba=target, stream=streams[f % 2].cuda_stream)
                    prev = tg2.submit(f, dev_unpack)
                    if prev is not None and rank == 0:
                        torch.cuda.current_stream().synchronize()
                        images.append(prev.cpu().numpy().copy())
            last = tg2.drain(dev_unpack)
            torch.cuda.synchronize()
            if rank == 0:
                images.append(last.cpu().numpy().copy())
            st = eng.frame_stats(streams[0].cuda_stream)
            steps = torch.tensor([float(st.accepted_steps)], dtype=torch.float64, device=dev)
            dist.all_reduce(steps)
            # the exchange in the compute pass's rgba16float format (bench.py --exchange rgba16f): the share is
            # rendered in f32, narrowed once into the half send buffer, gathered, de-interleaved 8 B per pixel
            eng.stats_accumulate(False)
            tg3 = D.TileGather(params, world, rank, 4, torch.float16, dev)
            scratch = torch.zeros(n_local, 4, dtype=torch.float32, device=dev)
            eng.render_frame_device(cam, rp, rgba=scratch, stream=torch.cuda.current_stream().cuda_stream)
            tg3.local_view(n_local).copy_(scratch)
            img16 = tg3.run(lambda rparams, r, packed, image: eng.unpack_tiles_device(
                rparams, r, packed, image, 8, torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            half_image = img16.float().cpu().numpy().copy() if rank == 0 else None
            if rank == 0:
                # the whole frame on this GPU, one rank
                whole = torch.zeros(w * h, 4, dtype=torch.float32, device=dev)
                eng.stats_accumulate(False)
                eng.render_frame_device(cam, params, rgba=whole, stream=torch.cuda.current_stream().cuda_stream)
                wst = eng.frame_stats()
                torch.cuda.synchronize()
                np.savez(out_path, whole=whole.cpu().numpy().reshape(h, w, 4), images=np.stack(images),
                         steps_sum=float(steps.item()), steps_whole=float(wst.accepted_steps),
                         half_image=half_image, whole_half=whole.half().float().cpu().numpy().reshape(h, w, 4))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h,arith", [(2, 200, 130, 0), (4, 512, 288, 1), (8, 960, 540, 1)])
def test_ranks_sharing_one_gpu_assemble_the_whole_frame_bitwise(tmp_path, engine_mod, world, w, h, arith):
    import torch.multiprocessing as mp
    out = str(tmp_path / "r.npz")
    mp.spawn(_worker, args=(world, _free_port(), w, h, arith, out), nprocs=world, join=True)
    r = np.load(out)
    assert r["images"].shape[0] == 5  # one synchronous frame + four pipelined ones
    for img in r["images"]:
        assert np.array_equal(img.view(np.uint32), r["whole"].view(np.uint32))
    assert r["steps_sum"] == 4 * r["steps_whole"]  # four accumulated frames, every ray counted once
    # rgba16f exchange: the whole frame rounded through binary16, bit for bit
    assert np.array_equal(r["half_image"].view(np.uint32), r["whole_half"].view(np.uint32))
    assert not np.array_equal(r["half_image"], r["whole"])


def _bench(world, *args, launcher=False):
    """bench.py's one-process-per-GPU host.  Default: `python bench.py --gpus N --launcher torchrun`
    (bench.py starts its own N ranks); launcher=True: under `python -m torch.distributed.run`, as
    the round-end driver starts N > 1.  (A bare `python bench.py --gpus N` goes through the C ABI's
    multi-GPU handle instead: tests/test_gpu_multi_native.py.)"""
    env = dict(os.environ, GRV_BENCH_BACKEND="gloo", GRV_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    if launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")]
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--launcher", "torchrun"]
    cmd += ["--gpus", str(world)] + list(args)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_self_launched_ranks_and_launcher_started_ranks_agree():
    """`python bench.py --gpus 2 --launcher torchrun` with no launcher around it must start two ranks
    and say so; the same command line under torch.distributed.run must give the same workload, and
    both lines carry the audit fields of a scaling point."""
    bare = _bench(2, "--width", "640", "--height", "360", "--steps", "5", "--warmup", "2")
    assert bare["n_gpus"] == 2 and bare["ranks"] == 2 and bare["rank_devices"] == [0, 0]
    under = _bench(2, "--width", "640", "--height", "360", "--steps", "5", "--warmup", "2", launcher=True)
    assert under["n_gpus"] == 2 and under["ranks"] == 2
    assert bare["config"]["accepted_steps_per_frame"] == under["config"]["accepted_steps_per_frame"]
    for ln in (bare, under):
        assert "torchrun" in ln["launcher"] and ln["transport"] == "gloo" and ln["rccl_version"] is None
        rk = ln["rank_integrate_ms"]
        assert len(rk["per_rank"]) == 2 and 0 < rk["min"] <= rk["max"]


def test_bench_refuses_more_ranks_than_devices():
    """Without the one-device test hook a --gpus 8 command on a smaller box must fail, loudly and
    with the device count -- never print a line for fewer GPUs than asked."""
    import torch
    have = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK",
                                                            "GRV_BENCH_ONE_DEVICE", "GRV_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 1), "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "only %d HIP device" % have in r.stderr
    # a launcher whose world size disagrees with --gpus is refused as well
    env2 = dict(env, GRV_BENCH_BACKEND="gloo", GRV_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env2)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.parametrize("world", [2, 8])
def test_bench_strong_split_runs_with_several_ranks(world):
    """bench.py's own N > 1 loop (strong split of the fixed frame, two frames in flight, pipelined
    gather, max-over-ranks timing, summed steps) against the one-rank run of the same frame."""
    one = _bench(1, "--width", "640", "--height", "360", "--steps", "4", "--warmup", "1", "--no-cpu-baseline")
    many = _bench(world, "--width", "640", "--height", "360", "--steps", "4", "--warmup", "1")
    assert many["n_gpus"] == world and many["scaling"] == "strong" and many["steps"] == 4
    assert many["config"]["rays"] == one["config"]["rays"] == 640 * 360
    assert many["config"]["accepted_steps_per_frame"] == one["config"]["accepted_steps_per_frame"]
    assert many["config"]["frames_in_flight"] == 2 and many["config"]["host_waits_in_frame_loop"] == 0
    assert "split over %d GPUs" % world in many["config"]["workload"]
    assert many["config"]["exchange"] == "rgba32f"
    half = _bench(world, "--width", "640", "--height", "360", "--steps", "4", "--warmup", "1", "--exchange", "rgba16f")
    assert half["config"]["exchange"] == "rgba16f" and half["n_gpus"] == world
    assert half["config"]["accepted_steps_per_frame"] == one["config"]["accepted_steps_per_frame"]
    assert many["value"] > 0 and many["roofline"]["avg_launch_ms"] > 0
    assert "cpu_baseline" not in many


def test_bench_config4_runs_with_several_ranks():
    one = _bench(1, "--config", "c4", "--width", "512", "--height", "288", "--steps", "3", "--warmup", "1",
                 "--no-cpu-baseline")
    many = _bench(2, "--config", "c4", "--width", "512", "--height", "288", "--steps", "3", "--warmup", "1")
    assert many["n_gpus"] == 2 and many["dtype"] == "f32" and many["config"]["arith"] == "packed"
    # the packed march pairs horizontally adjacent slots; a tile share pairs the same pixels
    assert many["config"]["accepted_steps_per_frame"] == one["config"]["accepted_steps_per_frame"]
    assert many["config"]["rays"] == 512 * 288
