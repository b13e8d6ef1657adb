"""gloo tests (world size 2, 3 and 4 incl. ranks that own no tile) of the N>1 path: tile partition -> one
gather to rank 0 -> de-interleave.  The per-rank renderer is replaced by a deterministic CPU fill so
the test checks exactly the distributed plumbing bench.py uses on GPUs."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _library_or_skip():
    """The tile deal and the parameter defaults live in libgravitas_hip.so (one implementation, include/gravitas_abi.h);
    the spawned workers would otherwise die with an opaque GravitasError each.  conftest's engine_mod builds the
    library when hipcc is there; where it cannot be had, say so once, here."""
    sys.path.insert(0, ROOT)
    import blackhole_simulation_amd as bh
    try:
        if not os.path.exists(bh.library_path()):
            bh.build_library()
        bh.load_library()
    except Exception as e:  # noqa: BLE001
        pytest.skip("libgravitas_hip.so cannot be built / loaded here (%s): the gloo workers need its tile deal" % e,
                    allow_module_level=True)


_library_or_skip()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, w, h, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import blackhole_simulation_amd as bh
    from blackhole_simulation_amd import distributed as D

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = bh.render_params(w, h)
        tiles = D.tiles_of_rank(w, h, world, rank)
        packed = torch.zeros((len(tiles) * 4096, 4), dtype=torch.float32)
        for tl, t in enumerate(tiles):
            x0, y0 = D.tile_origin(t, w, world)
            ys, xs = np.meshgrid(np.arange(64) + y0, np.arange(64) + x0, indexing="ij")
            blk = np.stack([xs, ys, xs * 0 + rank, xs * 0 + 1], -1).astype(np.float32)
            packed[tl * 4096:(tl + 1) * 4096] = torch.from_numpy(blk.reshape(-1, 4))
        img = D.gather_tiles(packed, p, world, rank, None, D.host_unpack)
        if rank == 0:
            np.save(out_path, img.numpy())
        else:
            assert img is None
        # pipelined form (bench.py's default for N > 1): frame i's gather is issued
        # asynchronously and completed when frame i+1 is submitted; three frames whose fourth
        # channel carries the frame number must come back in order and intact
        tg = D.TileGather(p, world, rank, 4, torch.float32, torch.device("cpu")).enable_pipeline()
        got = []
        for f in range(3):
            view = tg.pipelined_view(f, packed.shape[0])
            view.copy_(packed)
            view[:, 3] = float(f)
            prev = tg.submit(f, D.host_unpack)
            if prev is not None:
                got.append(prev.clone())
        last = tg.drain(D.host_unpack)
        if rank == 0:
            got.append(last.clone())
            assert len(got) == 3 and tg.drain(D.host_unpack) is None
            for f, im in enumerate(got):
                assert torch.equal(im[..., :3], img[..., :3]) and bool((im[..., 3] == float(f)).all())
        else:
            assert last is None and not got
        dist.barrier()
    finally:
        dist.destroy_process_group()


# world 3: an odd deal; world 4 on 100x60: two tiles in all, ranks 2 and 3 own none and still take part
@pytest.mark.parametrize("world,w,h", [(2, 200, 130), (2, 256, 128), (3, 200, 130), (4, 100, 60)])
def test_tile_gather_over_gloo_ranks(tmp_path, world, w, h):
    import torch.multiprocessing as mp
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, out), nprocs=world, join=True)
    img = np.load(out)
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    assert np.array_equal(img[..., 0], xs) and np.array_equal(img[..., 1], ys)
    assert np.all(img[..., 3] == 1.0)
    sys.path.insert(0, ROOT)
    from blackhole_simulation_amd import distributed as D
    owner = ((ys // 64) * D.tile_pitch(w, world) + xs // 64) % world
    assert np.array_equal(img[..., 2], owner)  # round-robin tile -> rank map


def test_single_rank_gather_is_the_identity():
    """world == 1: the renderer's output is already row-major (tile_world <= 1), so the
    'gather' must hand it back unpermuted -- with and without the collective walked."""
    import torch
    sys.path.insert(0, ROOT)
    import blackhole_simulation_amd as bh
    from blackhole_simulation_amd import distributed as D
    w, h = 200, 130
    p = bh.render_params(w, h)
    truth = torch.arange(w * h * 4, dtype=torch.float32).reshape(h * w, 4)
    img = D.gather_tiles(truth, p, 1, 0, None, D.host_unpack)
    assert torch.equal(img.reshape(-1, 4), truth)
    tg = D.TileGather(p, 1, 0, 4, torch.float32, torch.device("cpu")).enable_pipeline()
    tg.pipelined_view(0, w * h).copy_(truth)
    assert tg.submit(0, D.host_unpack) is None
    assert torch.equal(tg.drain(D.host_unpack).reshape(-1, 4), truth)


def _one_rank_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import blackhole_simulation_amd as bh
    from blackhole_simulation_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w, h = 130, 70
        p = bh.render_params(w, h)
        truth = torch.arange(w * h * 4, dtype=torch.float32).reshape(h * w, 4)
        tg = D.TileGather(p, 1, 0, 4, torch.float32, torch.device("cpu"))
        tg.local_view(w * h).copy_(truth)
        img = tg.run(D.host_unpack, force_collective=True)  # bench.py under a 1-rank torchrun
        np.save(out_path, (img.reshape(-1, 4) == truth).all().numpy())
    finally:
        dist.destroy_process_group()


def test_one_rank_collective_walk(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok.npy")
    mp.spawn(_one_rank_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    assert bool(np.load(out))
