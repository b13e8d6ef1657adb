"""Handle lifecycle on the GPU box: engines (and the multi-GPU handle with virtual ranks) created, used
on every kind of entry point and destroyed in a loop must give back what they took -- device memory
(hipMemGetInfo) and host memory (RSS).  The reference creates one PhysicsEngine per worker start
(src/workers/physics.worker.ts:60-64) and re-creates it on every INIT message; a handle that leaks
its workspaces would exhaust the device over a session."""
import numpy as np
import pytest


def _use(bh, torch, e, rgba, states):
    W, H = 256, 144
    th = np.deg2rad(97.0)
    cam = bh.camera_look_at((60.0 * np.sin(th), 60.0 * np.cos(th), 0.0), aspect=W / H)
    for arith in (bh.ARITH_FAST, bh.ARITH_STRICT):
        e.render_frame_device(cam, bh.render_params(W, H, arith=arith), rgba=rgba)
    e.render_frame_device(cam, bh.render_params(W, H, arith=bh.ARITH_FAST, segment_tries=16), rgba=rgba)
    e.frame_stats()
    e.integrate_batch(states, bh.engine.default_options(max_steps=300))
    e.integrate_ray_relativistic(states[0], 200, 1e-8, True)
    e.generate_spectrum_lut(128, 16, 1e5)
    e.generate_disk_lut()
    wp = bh.wgsl_params(W, H, cam, 1.0, 0.9, max_steps=64, arith=bh.ARITH_FAST_PACKED)
    e.render_frame_wgsl(wp, rgba)
    e.post_taa_resolve(W, H, rgba, rgba.clone(), torch.empty_like(rgba), arith=bh.ARITH_FAST)
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_engines_give_back_device_and_host_memory(engine_mod):
    import psutil
    import torch
    bh = engine_mod
    rng = np.random.default_rng(3)
    states = np.zeros((2000, 8))
    states[:, 1] = rng.uniform(8, 40, 2000)
    states[:, 2] = rng.uniform(0.3, 2.8, 2000)
    states[:, 4] = -1.0
    states[:, 5] = rng.uniform(-1, 0.2, 2000)
    states[:, 7] = rng.uniform(-5, 5, 2000)
    rgba = torch.zeros(256 * 144, 4, dtype=torch.float32, device="cuda:0")
    proc = psutil.Process()

    def cycle():
        with bh.PhysicsEngine(1.0, 0.9) as e:
            _use(bh, torch, e, rgba, states)
        with bh.MultiEngine(1.0, 0.9, virtual_ranks=4) as m:
            th = np.deg2rad(97.0)
            cam = bh.camera_look_at((60.0 * np.sin(th), 60.0 * np.cos(th), 0.0), aspect=256 / 144)
            img = rgba.view(144, 256, 4)
            for _ in range(3):
                m.render_frame_device(cam, bh.render_params(256, 144, arith=bh.ARITH_FAST), img)
            m.synchronize()

    for _ in range(3):  # runtime pools, code-object loads, the allocator's own caches
        cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    rss0 = proc.memory_info().rss
    n = 25
    for _ in range(n):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    rss1 = proc.memory_info().rss
    # one engine holds ~10 MB of device workspace for these sizes: a leak of it would show as n x that
    assert free0 - free1 < 16 * 2 ** 20, "device memory not returned: %.1f MB over %d cycles" % ((free0 - free1) / 2 ** 20, n)
    assert rss1 - rss0 < 48 * 2 ** 20, "host memory grew by %.1f MB over %d cycles" % ((rss1 - rss0) / 2 ** 20, n)


@pytest.mark.gpu
def test_frame_loop_resources_are_given_back(engine_mod):
    """The per-handle resources the frame-loop paths create lazily -- measured dispatch orders and their side stream,
    the compacting schedule's counters, feedback block and head-start events, the control stream with its workspace
    set, device images (own and shared compute streams, copy streams, pinned counter blocks) -- go with the handle."""
    import psutil
    import torch
    bh = engine_mod
    W, H = 1920, 1080                      # 2 073 600 rays: above the size from which one-launch frames take an order
    th = np.deg2rad(97.0)
    cam = bh.camera_look_at((20.0 * np.sin(th), 20.0 * np.cos(th), 0.0), aspect=W / H)
    one = np.array([[0.0, 20.0, 1.5, 0.0, -1.0, -0.9, 0.0, 3.0]])
    proc = psutil.Process()

    def cycle():
        with bh.PhysicsEngine(1.0, 0.999) as e:
            imgs = [e.create_image(W, H)]
            imgs.append(e.create_image(W, H, stream_of=imgs[0]))
            imgs.append(e.create_image(W, H))
            p = bh.render_params(W, H, arith=bh.ARITH_FAST, tolerance=1e-7, max_steps=400)
            k = bh.render_params(W, H, arith=bh.ARITH_FAST, tolerance=1e-7, max_steps=400, segment_tries=16)
            for i in range(6):
                e.render_frame_image(cam, p if i < 3 else k, imgs[i % 3])
                e.integrate_ray_relativistic(one[0], 50, 1e-8, True)      # the control stream beside queued frames
            gp = bh.glsl_params(W, H, 1.0, 0.999, max_ray_steps=64, arith=bh.ARITH_FAST)
            e.render_frame_glsl_image(gp, imgs[2])
            got = imgs[1].read()
            assert got.shape == (H, W, 4)
            st = imgs[0].stats()
            assert st.rays == W * H
            e.synchronize()
            for im in imgs:
                im.close()

    for _ in range(2):
        cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    rss0 = proc.memory_info().rss
    n = 8
    for _ in range(n):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    rss1 = proc.memory_info().rss
    # one cycle holds ~0.5 GB of device memory (three 33 MB images, two 1080p ray workspaces): a leak of any would show
    assert free0 - free1 < 24 * 2 ** 20, "device memory not returned: %.1f MB over %d cycles" % ((free0 - free1) / 2 ** 20, n)
    assert rss1 - rss0 < 64 * 2 ** 20, "host memory grew by %.1f MB over %d cycles" % ((rss1 - rss0) / 2 ** 20, n)
