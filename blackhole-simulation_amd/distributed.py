"""Image-plane partition across the GPUs of one node: one process per GPU, each
rank renders the 64x64 tiles k with k % world == rank (round-robin, so shadow and
photon-ring tiles spread evenly; tile grid as physics-engine/_legacy_src/tiling.rs:38-56;
tile ids use a row pitch coprime with the rank count so the deal shifts from row to row),
then ONE gather of the finished tiles to rank 0 (RCCL over xGMI when the backend
is "nccl") and a de-interleave on rank 0.  No other collective touches the path.

torch / torch.distributed are plumbing only (device buffers, process group).
"""
import ctypes as C

from . import engine as _eng

TILE = 64


def tile_pitch(width, world):
    """Row pitch of the tile ids: the first integer >= ceil(width / 64) coprime with `world`.  Ids in
    the pad column(s) hold no pixels.  The deal lives in the library (grv_tile_pitch,
    include/gravitas_abi.h); this module only asks."""
    return int(_eng.load_library().grv_tile_pitch(int(width), max(int(world), 1)))


def tiles_total(width, height, world=1):
    """Tile ids of the frame as `world` ranks number them (pad columns included): grv_tiles_total."""
    return int(_eng.load_library().grv_tiles_total(int(width), int(height), max(int(world), 1)))


def tiles_of_rank(width, height, world, rank):
    """Global tile ids rendered by `rank`, in its packed order: grv_tiles_of_rank."""
    lib = _eng.load_library()
    n = lib.grv_tiles_of_rank(int(width), int(height), int(world), int(rank), None, 0)
    ids = (C.c_uint32 * max(n, 1))()
    lib.grv_tiles_of_rank(int(width), int(height), int(world), int(rank), ids, n)
    return [int(ids[k]) for k in range(n)]


def tile_origin(tile, width, world):
    """Pixel (x0, y0) of a tile id; x0 >= width for an id in a pad column: grv_tile_origin."""
    x0, y0 = C.c_uint32(), C.c_uint32()
    _eng.load_library().grv_tile_origin(int(tile), int(width), max(int(world), 1), C.byref(x0), C.byref(y0))
    return int(x0.value), int(y0.value)


def max_tiles_per_rank(width, height, world):
    return int(_eng.load_library().grv_max_tiles_per_rank(int(width), int(height), max(int(world), 1)))


def rank_params(params, world, rank):
    """Copy of `params` restricted to this rank's tiles."""
    p = _eng.RenderParams()
    C.memmove(C.byref(p), C.byref(params), C.sizeof(p))
    p.tile_world = world
    p.tile_rank = rank
    return p


class TileGather:
    """Pre-allocated buffers for the per-frame gather: the padded send buffer, rank 0's
    receive slots and the assembled image are created once, outside the frame loop."""

    def __init__(self, params, world, rank, channels, dtype, device, group=None):
        import torch
        self.params, self.world, self.rank, self.group = params, world, rank, group
        self.n_max = max_tiles_per_rank(params.width, params.height, world) * TILE * TILE
        self.send = torch.zeros((self.n_max, channels), dtype=dtype, device=device)
        self.parts = None
        self.image = None
        if rank == 0:
            self.parts = [torch.empty((self.n_max, channels), dtype=dtype, device=device)
                          for _ in range(world)]
            self.image = torch.zeros((params.height, params.width, channels), dtype=dtype, device=device)
        self.rparams = [rank_params(params, world, r) for r in range(world)]

    # ---- where a rank's time goes when it is not computing: every point at which the rank's stream is made to
    # wait for the exchange (the gather of the previous frame, the reader of a send buffer) can be bracketed by
    # two events on that stream; exchange_wait_ms() resolves them after the loop.  Off unless asked for. ----
    def enable_wait_timing(self, enable=True):
        self._wait_events = [] if enable else None
        return self

    def _timed_wait(self, wait):
        ev = getattr(self, "_wait_events", None)
        if ev is None:
            return wait()
        import torch
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = wait()
        b.record()
        ev.append((a, b))
        return out

    def exchange_wait_ms(self, reset=True):
        """Total time (ms) this rank's streams were held at the exchange's wait points since the last reset
        (call after a device synchronise); 0.0 when timing is off or nothing was waited for."""
        ev = getattr(self, "_wait_events", None)
        if not ev:
            return 0.0
        total = sum(a.elapsed_time(b) for a, b in ev)
        if reset:
            del ev[:]
        return float(total)

    def local_view(self, n_local):
        """Render directly into the (padded) send buffer: no staging copy."""
        return self.send[:n_local]

    def run(self, unpack, force_collective=False):
        """Synchronous form: gather this frame's tiles now, de-interleave on rank 0."""
        return self._unpack(unpack, self._timed_wait(lambda: self._gather(self.send, force_collective, False)), self.send)

    # ---- pipelined form: the gather of frame i runs while frame i+1 is integrated ----------
    # Two send buffers alternate; the exchange is issued asynchronously (RCCL runs it on its own
    # stream over xGMI) and waited for only after the next frame's kernels are in the queue, so
    # the only exposed communication is the last frame's.  The data path still holds exactly one
    # collective per frame.  The caller may put even and odd frames on two different streams (two
    # frames in flight): every wait below is a wait of the CURRENT stream, and a send buffer is
    # handed out again only behind the gather that last read it.
    def enable_pipeline(self):
        import torch
        if getattr(self, "sends", None) is None:
            self.sends = [self.send, torch.zeros_like(self.send)]
            self._pending = None  # (work, send buffer, collective?)
            self._reader = [None, None]  # per send buffer: the gather that last read it
        return self

    def pipelined_view(self, frame_index, n_local):
        """Render target for `frame_index` (no staging copy).  Orders the current stream behind
        the gather of frame_index - 2, which read the same buffer."""
        b = frame_index % 2
        if self._reader[b] is not None:
            self._timed_wait(self._reader[b].wait)
            self._reader[b] = None
        return self.sends[b][:n_local]

    def submit(self, frame_index, unpack, force_collective=False):
        """Call after frame `frame_index` was rendered into pipelined_view(frame_index): completes
        the previous frame's exchange (its gather overlapped this frame's integration), then
        starts this frame's.  Returns the previous frame's image on rank 0 (None on the first
        call and on other ranks)."""
        img = self.drain(unpack)
        buf = self.sends[frame_index % 2]
        self._pending = (self._gather(buf, force_collective, True), buf)
        self._reader[frame_index % 2] = self._pending[0][0]
        return img

    def drain(self, unpack):
        """Finish the outstanding exchange, if any."""
        if self._pending is None:
            return None
        (work, collective), buf = self._pending
        self._pending = None
        if work is not None:
            self._timed_wait(work.wait)
        return self._unpack(unpack, (None, collective), buf)

    def _gather(self, buf, force_collective, async_op):
        import torch.distributed as dist
        collective = self.world > 1 or force_collective  # force: 1-rank dry run of the RCCL call
        work = None
        if collective:
            work = dist.gather(buf, self.parts if self.rank == 0 else None, dst=0, group=self.group,
                               async_op=async_op)
        return work, collective

    def _unpack(self, unpack, gathered, buf):
        _, collective = gathered
        if self.rank != 0:
            return None
        parts = self.parts if collective else [buf]
        if self.world <= 1:
            # a whole-frame render (tile_world <= 1) is already row-major (gravitas_abi.h,
            # GrvFrameBuffers): nothing to de-interleave, the image is the first H*W pixels
            n = self.params.height * self.params.width
            self.image.view(n, -1).copy_(parts[0][:n])
            return self.image
        for r in range(self.world):
            unpack(self.rparams[r], r, parts[r], self.image)
        return self.image


def gather_tiles(local_packed, params, world, rank, group=None, unpack=None):
    """One-shot form of TileGather (allocates per call): gather every rank's packed tiles on
    rank 0 and de-interleave them.

    local_packed: tensor [n_tiles_local * 4096, C] (this rank's tiles, tile order).
    unpack(rank_params, r, packed_tensor, image_tensor): scatters rank r's tiles into
    the row-major image (device kernel or host memcpy, chosen by the caller).
    Returns the [H, W, C] image on rank 0, None elsewhere.
    """
    tg = TileGather(params, world, rank, local_packed.shape[1], local_packed.dtype,
                    local_packed.device, group)
    tg.send[: local_packed.shape[0]] = local_packed
    return tg.run(unpack)


def host_unpack(rparams, r, packed, image):
    """CPU-tensor de-interleave through the C ABI's host helper."""
    rc = _eng.load_library().grv_unpack_tiles(
        C.byref(rparams), r, C.c_void_p(packed.data_ptr()), C.c_void_p(image.data_ptr()),
        packed.shape[1] * packed.element_size())
    if rc != 0:
        raise _eng.GravitasError("grv_unpack_tiles failed: %d" % rc)


def render_frame_distributed(eng, camera, params, group=None, stream=None):
    """Render this rank's tiles on its GPU, gather on rank 0.  Returns (image|None, FrameStats)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    p = rank_params(params, world, rank)
    n = eng.frame_ray_count(p)
    rgba = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    eng.render_frame_device(camera, p, rgba=rgba, stream=stream)
    st = eng.frame_stats(stream)

    def dev_unpack(rp, r, packed, image):
        eng.unpack_tiles_device(rp, r, packed, image, packed.shape[1] * packed.element_size(), stream)

    image = gather_tiles(rgba, params, world, rank, group, dev_unpack)
    return image, st
