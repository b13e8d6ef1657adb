// engine_multi.hip -- the image plane across the GPUs of one node, behind the C ABI.
//
// One process, one host thread and two streams per device (single-process-multi-GPU
// suffices inside a node).  Rank k owns one grv_engine on its device and renders the 64x64 tiles
// t with t % G == k (physics-engine/_legacy_src/tiling.rs:38-56 row-major grid, dealt round-robin;
// the kernels take tile_world / tile_rank) straight into a packed send buffer.  The ONE exchange
// per frame is the gather of the finished tiles to rank 0:
//   GRV_TRANSPORT_RCCL      one ncclGroupStart ... ncclSend (every rank >= 1, on its render stream)
//                           / ncclRecv x (G-1) (rank 0) ... ncclGroupEnd over xGMI: G-1 concurrent
//                           point-to-point transfers, one link each, no ring;
//   GRV_TRANSPORT_PEER_COPY every rank pushes its tiles into rank 0's receive slot with
//                           hipMemcpyPeerAsync on its render stream (also the transport of G
//                           VIRTUAL ranks on one device, which RCCL refuses: the assembly logic is
//                           testable on a one-GPU box).
// Rank 0 then de-interleaves every rank's slot into the caller's row-major image
// (unpack_tiles16_kernel).  Even and odd frames use separate streams, send buffers and receive
// slots: frame i+1's kernels are queued while frame i's tail, exchange and unpack still run, and
// the call never waits for the device.
#include "engine_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

using namespace grvhost;

namespace {

// RCCL is bound at run time (dlopen) when a handle asks for that transport: hosts that only use
// one GPU, or the peer-copy transport, need no librccl at all.
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    std::string err;
    bool ready = false;
    // Binds librccl once.  A failed attempt leaves the object exactly as it was (no half-bound
    // library: a later load() must not report success over null entry points).
    bool load() {
        if (ready) return true;
        void *h = nullptr;
        std::string why;
        // GRV_RCCL_LIBRARY names the library of a non-standard install; when set it is the only
        // candidate (a wrong path fails loudly instead of binding some other copy)
        const char *forced = std::getenv("GRV_RCCL_LIBRARY");
        std::vector<const char *> names;
        if (forced && *forced) names = {forced};
        else names = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *name : names) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
            const char *m = dlerror(); // returns the message ONCE and resets it: read it into a variable
            if (why.empty()) why = m ? m : "not found";
        }
        if (!h) {
            err = std::string("dlopen(") + names[0] + "): " + why;
            return false;
        }
        const char *missing = nullptr;
        auto sym = [&](const char *n) {
            void *p = dlsym(h, n);
            if (!p && !missing) missing = n;
            return p;
        };
        auto init_all = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        auto destroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        auto gstart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        auto gend = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        auto send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        auto recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        auto estr = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        auto ver = reinterpret_cast<decltype(GetVersion)>(sym("ncclGetVersion"));
        if (missing) {
            err = std::string("librccl lacks ") + missing;
            dlclose(h);
            return false;
        }
        lib = h;
        CommInitAll = init_all;
        CommDestroy = destroy;
        GroupStart = gstart;
        GroupEnd = gend;
        Send = send;
        Recv = recv;
        GetErrorString = estr;
        GetVersion = ver;
        ready = true;
        return true;
    }
};
RcclApi g_rccl;
std::mutex g_rccl_mu;
// why the last grv_engine_create_multi* of this thread failed (no handle exists to carry it)
thread_local std::string g_create_err;
int cfail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_create_err = buf;
    std::fprintf(stderr, "gravitas: %s\n", buf);
    return code;
}

// One worker thread per rank.  run(job) hands `job(rank)` to every worker and returns when all of
// them have finished QUEUEING (the device work stays asynchronous); the first non-zero status wins.
class RankThreads {
  public:
    explicit RankThreads(int n) : n_(n), rc_(n, 0) {
        for (int r = 0; r < n; ++r) th_.emplace_back([this, r] { loop(r); });
    }
    ~RankThreads() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    int run(const std::function<int(int)> &job) {
        std::unique_lock<std::mutex> lk(mu_);
        job_ = &job;
        pending_ = n_;
        ++gen_;
        cv_.notify_all();
        done_.wait(lk, [this] { return pending_ == 0; });
        job_ = nullptr;
        for (int rc : rc_)
            if (rc != 0) return rc;
        return 0;
    }

  private:
    void loop(int r) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<int(int)> *job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                job = job_;
            }
            const int rc = (*job)(r);
            {
                std::lock_guard<std::mutex> lk(mu_);
                rc_[r] = rc;
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    int n_;
    std::vector<std::thread> th_;
    std::vector<int> rc_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<int(int)> *job_ = nullptr;
    int pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

} // namespace

struct grv_multi {
    int G = 0;
    int transport = GRV_TRANSPORT_PEER_COPY;
    bool virtual_ranks = false;
    bool self_exchange = false; // test hook: rank 0's own share travels through the transport too
    std::vector<int> dev;
    std::vector<grv_engine *> eng;
    std::string err;

    int format = GRV_EXCHANGE_RGBA32F; // what travels: RGBA f32 (16 B / pixel) or RGBA binary16 (8 B / pixel)

    struct Rank {
        hipStream_t s[2] = {nullptr, nullptr};
        float *send[2] = {nullptr, nullptr}; // packed tile-order RGBA f32 (ranks >= 1; rank 0 under self_exchange
                                             // and, RGBA16F, as its f32 render target)
        void *send16[2] = {nullptr, nullptr}; // RGBA16F: the share as it travels (ranks that exchange)
        hipEvent_t arrived[2] = {nullptr, nullptr}; // this rank's tiles of frame parity b sit in rank 0's slot
    };
    std::vector<Rank> rank;
    size_t slot_px = 0; // pixels per receive slot / send buffer (max tiles of a rank * 4096)

    // rank 0 side
    float *recv[2] = {nullptr, nullptr}; // [G][slot_px][4] per parity (RGBA16F: [G][slot_px] x 8 B in the same allocation)
    hipStream_t rs[2] = {nullptr, nullptr}; // exchange + unpack streams
    hipEvent_t unpacked[2] = {nullptr, nullptr};
    bool unpacked_rec[2] = {false, false};
    hipEvent_t caller_ready = nullptr;
    float *image = nullptr; // host-pointer entry: assembled image on rank 0's device
    size_t image_px = 0;

    std::vector<ncclComm_t> comm;
    RankThreads *threads = nullptr;
    uint64_t frame = 0;
};

namespace {

int mfail(grv_multi *m, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (m) m->err = buf;
    return code;
}

#define GRVM_HIP(m, call)                                                                    \
    do {                                                                                     \
        hipError_t _st = (call);                                                             \
        if (_st != hipSuccess)                                                               \
            return mfail((m), _st == hipErrorOutOfMemory ? GRV_ERR_OOM : GRV_ERR_HIP,         \
                         "%s failed: %s", #call, hipGetErrorString(_st));                    \
    } while (0)
#define GRVM_NCCL(m, call)                                                                   \
    do {                                                                                     \
        ncclResult_t _st = (call);                                                           \
        if (_st != ncclSuccess)                                                              \
            return mfail((m), GRV_ERR_HIP, "%s failed: %s", #call, g_rccl.GetErrorString(_st)); \
    } while (0)

size_t max_tiles_per_rank(uint32_t width, uint32_t height, uint32_t world) {
    const size_t total = (size_t)tile_pitch(width, world) * ((height + 63u) / 64u);
    return (total + world - 1) / world;
}

// Waits for the devices and frees every exchange buffer (the next frame allocates for the mode /
// size then in force).
int drop_buffers(grv_multi *m) {
    for (int r = 0; r < m->G; ++r) {
        GRVM_HIP(m, hipSetDevice(m->dev[r]));
        GRVM_HIP(m, hipDeviceSynchronize());
        for (int b = 0; b < 2; ++b) {
            if (m->rank[r].send[b]) (void)hipFree(m->rank[r].send[b]);
            if (m->rank[r].send16[b]) (void)hipFree(m->rank[r].send16[b]);
            m->rank[r].send[b] = nullptr;
            m->rank[r].send16[b] = nullptr;
        }
    }
    GRVM_HIP(m, hipSetDevice(m->dev[0]));
    for (int b = 0; b < 2; ++b) {
        if (m->recv[b]) (void)hipFree(m->recv[b]);
        m->recv[b] = nullptr;
        m->unpacked_rec[b] = false;
    }
    m->slot_px = 0;
    return GRV_OK;
}

// Buffers sized for a width x height frame split over G ranks (grown on demand; growing waits for
// the device -- frames of a fixed size never do).
int ensure_buffers(grv_multi *m, uint32_t width, uint32_t height) {
    const size_t need = max_tiles_per_rank(width, height, (uint32_t)m->G) * 4096u;
    if (need <= m->slot_px) return GRV_OK;
    int rc = drop_buffers(m);
    if (rc != GRV_OK) return rc;
    const bool half = m->format == GRV_EXCHANGE_RGBA16F;
    const size_t wire = half ? 8u : 16u; // bytes per pixel as exchanged
    for (int b = 0; b < 2; ++b)
        GRVM_HIP(m, hipMalloc(reinterpret_cast<void **>(&m->recv[b]), (size_t)m->G * need * wire));
    for (int r = 0; r < m->G; ++r) {
        const bool direct = (r == 0 && !m->self_exchange); // rank 0's share needs no transport
        if (direct && !half) continue;                     // ... and, RGBA f32, is rendered into its receive slot
        GRVM_HIP(m, hipSetDevice(m->dev[r]));
        for (int b = 0; b < 2; ++b) {
            GRVM_HIP(m, hipMalloc(reinterpret_cast<void **>(&m->rank[r].send[b]), need * 16u));
            if (half && !direct) GRVM_HIP(m, hipMalloc(&m->rank[r].send16[b], need * 8u));
        }
    }
    m->slot_px = need;
    return GRV_OK;
}

// The frame skeleton shared by the f64 frame and the f32 compute march.
//   render(rank, engine, target, stream): queue this rank's tile share into `target` (packed tile
//   order, RGBA f32) on `stream`; n_px(rank): pixels of that share.
int run_frame(grv_multi *m, uint32_t width, uint32_t height, float *d_rgba, hipStream_t caller,
              const std::function<int(int, grv_engine *, float *, hipStream_t)> &render) {
    if (!d_rgba) return mfail(m, GRV_ERR_INVALID, "null image");
    if (width == 0 || height == 0) return mfail(m, GRV_ERR_INVALID, "empty frame");
    const int G = m->G;
    const bool half = m->format == GRV_EXCHANGE_RGBA16F;
    if (G == 1 && !m->self_exchange) {
        // one rank: the whole frame is already row-major (GrvFrameBuffers), nothing to exchange
        GRVM_HIP(m, hipSetDevice(m->dev[0]));
        const int rc = render(0, m->eng[0], d_rgba, caller);
        if (rc != GRV_OK) return mfail(m, rc, "rank 0: %s", grv_last_error(m->eng[0]));
        // RGBA16F: the image a G-rank handle assembles is the half-rounded frame; so is this one
        if (half) GRVM_HIP(m, launch_post_quantize(d_rgba, width * height, caller));
        m->frame++;
        return GRV_OK;
    }
    int rc = ensure_buffers(m, width, height);
    if (rc != GRV_OK) return rc;
    const int b = (int)(m->frame & 1u);
    GrvRenderParams geom{};
    geom.width = width;
    geom.height = height;
    geom.tile_world = (uint32_t)G;
    std::vector<size_t> n_px(G);
    for (int r = 0; r < G; ++r) {
        geom.tile_rank = (uint32_t)r;
        n_px[r] = grv_frame_ray_count(&geom);
    }
    const bool rccl = m->transport == GRV_TRANSPORT_RCCL;
    const size_t wire = half ? 8u : 16u;

    // every rank: render its share (and, peer-copy transport, push it to rank 0) on its own thread
    rc = m->threads->run([&](int r) -> int {
        grv_multi::Rank &R = m->rank[r];
        if (hipSetDevice(m->dev[r]) != hipSuccess) return GRV_ERR_HIP;
        hipStream_t s = R.s[b];
        // the receive slot of this parity is free once frame - 2 has been unpacked
        if (m->unpacked_rec[b] && hipStreamWaitEvent(s, m->unpacked[b], 0) != hipSuccess) return GRV_ERR_HIP;
        // receive slot of rank r: f32 pixels, or (RGBA16F) 8-byte pixels in the same allocation
        char *slot = reinterpret_cast<char *>(m->recv[b]) + (size_t)r * m->slot_px * wire;
        const bool direct = (r == 0 && !m->self_exchange);
        float *target = (direct && !half) ? reinterpret_cast<float *>(slot) : R.send[b];
        if (n_px[r] > 0) {
            const int st = render(r, m->eng[r], target, s);
            if (st != GRV_OK) return st;
        }
        const void *wire_src = target;
        if (half && n_px[r] > 0) {
            // narrow the share to binary16 where it was rendered: rank 0's straight into its slot
            void *dst = direct ? static_cast<void *>(slot) : R.send16[b];
            if (launch_pack_half(target, dst, n_px[r], s) != hipSuccess) return GRV_ERR_HIP;
            wire_src = dst;
        }
        if (!direct && !rccl && n_px[r] > 0) {
            const hipError_t st = (m->dev[r] == m->dev[0])
                                      ? hipMemcpyAsync(slot, wire_src, n_px[r] * wire, hipMemcpyDeviceToDevice, s)
                                      : hipMemcpyPeerAsync(slot, m->dev[0], wire_src, m->dev[r], n_px[r] * wire, s);
            if (st != hipSuccess) return GRV_ERR_HIP;
        }
        // RCCL: the event marks "rendered"; the transfer itself is ordered by ncclRecv on rank 0's stream
        if (hipEventRecord(R.arrived[b], s) != hipSuccess) return GRV_ERR_HIP;
        return GRV_OK;
    });
    if (rc != GRV_OK) {
        for (int r = 0; r < G; ++r)
            if (m->eng[r] && *grv_last_error(m->eng[r])) return mfail(m, rc, "rank %d: %s", r, grv_last_error(m->eng[r]));
        return mfail(m, rc, "a rank failed to queue its share: %s", hipGetErrorString(hipGetLastError()));
    }

    // rank 0: the one exchange, then the de-interleave into the caller's image
    GRVM_HIP(m, hipSetDevice(m->dev[0]));
    hipStream_t rs = m->rs[b];
    if (rccl) {
        // one group: G-1 sends on the ranks' render streams (behind their kernels), G-1 receives
        // on rank 0's exchange stream -- concurrent point-to-point transfers, one xGMI link each
        GRVM_NCCL(m, g_rccl.GroupStart());
        ncclResult_t gst = ncclSuccess; // a failed call must not leave the group open: always reach GroupEnd
        for (int r = (m->self_exchange ? 0 : 1); r < G && gst == ncclSuccess; ++r) {
            if (n_px[r] == 0) continue;
            // four channels per pixel either way: ncclFloat or ncclHalf elements
            const ncclDataType_t ty = half ? ncclHalf : ncclFloat;
            const void *src = half ? m->rank[r].send16[b] : static_cast<const void *>(m->rank[r].send[b]);
            gst = g_rccl.Send(src, n_px[r] * 4u, ty, 0, m->comm[r], m->rank[r].s[b]);
            if (gst == ncclSuccess)
                gst = g_rccl.Recv(reinterpret_cast<char *>(m->recv[b]) + (size_t)r * m->slot_px * wire, n_px[r] * 4u, ty, r,
                                  m->comm[0], rs);
        }
        const ncclResult_t gend = g_rccl.GroupEnd();
        if (gst != ncclSuccess) return mfail(m, GRV_ERR_HIP, "ncclSend / ncclRecv failed: %s", g_rccl.GetErrorString(gst));
        GRVM_NCCL(m, gend);
        if (!m->self_exchange) GRVM_HIP(m, hipStreamWaitEvent(rs, m->rank[0].arrived[b], 0));
    } else {
        for (int r = 0; r < G; ++r) GRVM_HIP(m, hipStreamWaitEvent(rs, m->rank[r].arrived[b], 0));
    }
    // the caller's image may still be read by work the caller queued earlier
    GRVM_HIP(m, hipEventRecord(m->caller_ready, caller));
    GRVM_HIP(m, hipStreamWaitEvent(rs, m->caller_ready, 0));
    if (G == 1) {
        // (self-exchange walk) one rank's share is the whole frame, already row-major
        if (half) GRVM_HIP(m, launch_widen_half(m->recv[b], d_rgba, n_px[0], rs));
        else GRVM_HIP(m, hipMemcpyAsync(d_rgba, m->recv[b], n_px[0] * 16u, hipMemcpyDeviceToDevice, rs));
    } else {
        for (int r = 0; r < G; ++r) {
            if (n_px[r] == 0) continue;
            geom.tile_rank = (uint32_t)r;
            FrameGeom FG;
            frame_geometry(geom, FG);
            const char *slot = reinterpret_cast<const char *>(m->recv[b]) + (size_t)r * m->slot_px * wire;
            if (half) GRVM_HIP(m, launch_unpack_tiles_half(FG, slot, d_rgba, rs));
            else GRVM_HIP(m, launch_unpack_tiles(FG, slot, d_rgba, 4u, rs));
        }
    }
    GRVM_HIP(m, hipEventRecord(m->unpacked[b], rs));
    m->unpacked_rec[b] = true;
    GRVM_HIP(m, hipStreamWaitEvent(caller, m->unpacked[b], 0)); // the image is complete in the caller's stream order
    m->frame++;
    return GRV_OK;
}

int create_common(double mass, double spin, const std::vector<int> &devs, bool virt, int transport,
                  grv_multi **out) {
    if (!out) return GRV_ERR_INVALID;
    *out = nullptr;
    g_create_err.clear();
    const int G = (int)devs.size();
    if (G < 1 || G > 64) return cfail(GRV_ERR_INVALID, "multi-GPU handle: %d ranks (1..64)", G);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return cfail(GRV_ERR_NO_DEVICE, "multi-GPU handle: no HIP device");
    }
    for (int d : devs)
        if (d < 0 || d >= count)
            return cfail(GRV_ERR_NO_DEVICE, "multi-GPU handle: device %d asked for, %d visible", d, count);
    if (transport == GRV_TRANSPORT_AUTO) transport = (virt || G == 1) ? GRV_TRANSPORT_PEER_COPY : GRV_TRANSPORT_RCCL;
    if (transport != GRV_TRANSPORT_RCCL && transport != GRV_TRANSPORT_PEER_COPY)
        return cfail(GRV_ERR_INVALID, "multi-GPU handle: unknown transport %d", transport);
    if (transport == GRV_TRANSPORT_RCCL && virt && G > 1) // RCCL: one rank per device
        return cfail(GRV_ERR_INVALID, "multi-GPU handle: RCCL takes one rank per device");
    grv_multi *m = new (std::nothrow) grv_multi();
    if (!m) return cfail(GRV_ERR_OOM, "multi-GPU handle: out of host memory");
    m->G = G;
    m->dev = devs;
    m->virtual_ranks = virt;
    m->transport = transport;
    m->eng.assign(G, nullptr);
    m->rank.resize(G);
    auto bail = [&](int code) {
        grv_multi_destroy(m);
        return code;
    };
    // a failed HIP call: its text goes to grv_multi_create_error (no failure leaves the reason empty)
    hipError_t hst = hipSuccess;
    auto hip_bail = [&](const char *what, int r) {
        const int code = cfail(hst == hipErrorOutOfMemory ? GRV_ERR_OOM : GRV_ERR_HIP, "multi-GPU handle: %s (rank %d, device %d): %s",
                               what, r, devs[r < 0 ? 0 : r], hipGetErrorString(hst));
        (void)hipGetLastError();
        return bail(code);
    };
#define GRVC_HIP(call, what, r)                      \
    do {                                             \
        hst = (call);                                \
        if (hst != hipSuccess) return hip_bail(what, r); \
    } while (0)
    for (int r = 0; r < G; ++r) {
        const int rc = grv_engine_create(mass, spin, devs[r], &m->eng[r]);
        if (rc != GRV_OK)
            return bail(cfail(rc, "multi-GPU handle: grv_engine_create on device %d (rank %d) failed with status %d%s",
                              devs[r], r, rc, rc == GRV_ERR_NO_DEVICE ? " (no usable HIP device)" : ""));
        GRVC_HIP(hipSetDevice(devs[r]), "hipSetDevice", r);
        for (int b = 0; b < 2; ++b) {
            GRVC_HIP(hipStreamCreateWithFlags(&m->rank[r].s[b], hipStreamNonBlocking), "hipStreamCreate", r);
            GRVC_HIP(hipEventCreateWithFlags(&m->rank[r].arrived[b], hipEventDisableTiming), "hipEventCreate", r);
        }
        // peer access rank r -> rank 0 for the push copies (RCCL sets up its own mappings)
        if (r > 0 && devs[r] != devs[0]) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devs[r], devs[0]) == hipSuccess && can) {
                hst = hipDeviceEnablePeerAccess(devs[0], 0);
                if (hst != hipSuccess && hst != hipErrorPeerAccessAlreadyEnabled) return hip_bail("hipDeviceEnablePeerAccess", r);
                (void)hipGetLastError();
            }
        }
    }
    GRVC_HIP(hipSetDevice(devs[0]), "hipSetDevice", 0);
    for (int b = 0; b < 2; ++b) {
        GRVC_HIP(hipStreamCreateWithFlags(&m->rs[b], hipStreamNonBlocking), "hipStreamCreate (exchange)", 0);
        GRVC_HIP(hipEventCreateWithFlags(&m->unpacked[b], hipEventDisableTiming), "hipEventCreate (exchange)", 0);
    }
    GRVC_HIP(hipEventCreateWithFlags(&m->caller_ready, hipEventDisableTiming), "hipEventCreate (caller)", 0);
#undef GRVC_HIP
    if (transport == GRV_TRANSPORT_RCCL) {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        // never a silent fall back to peer copies: a handle that asked for RCCL (AUTO between real
        // devices included) and cannot have it is refused, with the reason
        if (!g_rccl.load())
            return bail(cfail(GRV_ERR_NO_DEVICE, "RCCL transport unavailable: %s", g_rccl.err.c_str()));
        m->comm.assign(G, nullptr);
        const ncclResult_t st = g_rccl.CommInitAll(m->comm.data(), G, devs.data());
        if (st != ncclSuccess) {
            m->comm.clear();
            return bail(cfail(GRV_ERR_HIP, "ncclCommInitAll over %d devices failed: %s", G, g_rccl.GetErrorString(st)));
        }
    }
    m->threads = new (std::nothrow) RankThreads(G);
    if (!m->threads) return bail(cfail(GRV_ERR_OOM, "multi-GPU handle: out of host memory (rank threads)"));
    *out = m;
    return GRV_OK;
}

} // namespace

extern "C" {

int grv_engine_create_multi(double mass, double spin, uint64_t device_mask, int transport, grv_multi **out) {
    std::vector<int> devs;
    for (int d = 0; d < 64; ++d)
        if (device_mask >> d & 1u) devs.push_back(d);
    if (devs.empty()) return GRV_ERR_INVALID;
    return create_common(mass, spin, devs, false, transport, out);
}

int grv_engine_create_multi_virtual(double mass, double spin, int device, int ranks, grv_multi **out) {
    if (ranks < 1 || ranks > 64) return GRV_ERR_INVALID;
    return create_common(mass, spin, std::vector<int>(ranks, device), true, GRV_TRANSPORT_PEER_COPY, out);
}

void grv_multi_destroy(grv_multi *m) {
    if (!m) return;
    delete m->threads; // joins the workers
    m->threads = nullptr;
    for (int r = 0; r < m->G; ++r) {
        if (hipSetDevice(m->dev[r]) != hipSuccess) continue;
        (void)hipDeviceSynchronize();
    }
    if (!m->comm.empty())
        for (auto c : m->comm)
            if (c) (void)g_rccl.CommDestroy(c);
    for (int r = 0; r < m->G; ++r) {
        (void)hipSetDevice(m->dev[r]);
        for (int b = 0; b < 2; ++b) {
            if (m->rank[r].send[b]) (void)hipFree(m->rank[r].send[b]);
            if (m->rank[r].send16[b]) (void)hipFree(m->rank[r].send16[b]);
            if (m->rank[r].s[b]) (void)hipStreamDestroy(m->rank[r].s[b]);
            if (m->rank[r].arrived[b]) (void)hipEventDestroy(m->rank[r].arrived[b]);
        }
        if (m->eng[r]) grv_engine_destroy(m->eng[r]);
    }
    (void)hipSetDevice(m->dev[0]);
    for (int b = 0; b < 2; ++b) {
        if (m->recv[b]) (void)hipFree(m->recv[b]);
        if (m->rs[b]) (void)hipStreamDestroy(m->rs[b]);
        if (m->unpacked[b]) (void)hipEventDestroy(m->unpacked[b]);
    }
    if (m->caller_ready) (void)hipEventDestroy(m->caller_ready);
    if (m->image) (void)hipFree(m->image);
    delete m;
}

const char *grv_multi_last_error(const grv_multi *m) { return m ? m->err.c_str() : "null handle"; }
int grv_multi_rank_count(const grv_multi *m) { return m ? m->G : 0; }
int grv_multi_rank_device(const grv_multi *m, int rank) {
    return (m && rank >= 0 && rank < m->G) ? m->dev[rank] : -1;
}
int grv_multi_transport(const grv_multi *m) { return m ? m->transport : -1; }

const char *grv_multi_create_error(void) { return g_create_err.c_str(); }

int grv_rccl_probe(int *version, char *msg, size_t msg_len) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (version) *version = 0;
    if (msg && msg_len) msg[0] = 0;
    if (!g_rccl.load()) {
        if (msg && msg_len) std::snprintf(msg, msg_len, "%s", g_rccl.err.c_str());
        return GRV_ERR_NO_DEVICE;
    }
    int v = 0;
    const ncclResult_t st = g_rccl.GetVersion(&v);
    if (st != ncclSuccess) {
        if (msg && msg_len) std::snprintf(msg, msg_len, "ncclGetVersion: %s", g_rccl.GetErrorString(st));
        return GRV_ERR_HIP;
    }
    if (version) *version = v;
    return GRV_OK;
}

int grv_multi_test_self_exchange(grv_multi *m, int enable) {
    if (!m) return GRV_ERR_INVALID;
    if (!grv_test_hooks_unlocked())
        return mfail(m, GRV_ERR_INVALID, "grv_multi_test_self_exchange: verification hooks are locked (grv_test_hooks_unlock)");
    int rc = grv_multi_synchronize(m);
    if (rc != GRV_OK) return rc;
    // the buffers are laid out for the other mode: drop them, the next frame allocates afresh
    rc = drop_buffers(m);
    if (rc != GRV_OK) return rc;
    m->self_exchange = enable != 0;
    return GRV_OK;
}

int grv_multi_set_exchange_format(grv_multi *m, int format) {
    if (!m) return GRV_ERR_INVALID;
    if (format != GRV_EXCHANGE_RGBA32F && format != GRV_EXCHANGE_RGBA16F)
        return mfail(m, GRV_ERR_INVALID, "unknown exchange format %d", format);
    if (format == m->format) return GRV_OK;
    int rc = grv_multi_synchronize(m);
    if (rc != GRV_OK) return rc;
    rc = drop_buffers(m); // sized for the other pixel width
    if (rc != GRV_OK) return rc;
    m->format = format;
    return GRV_OK;
}
int grv_multi_exchange_format(const grv_multi *m) { return m ? m->format : -1; }
size_t grv_multi_exchange_bytes_per_frame(const grv_multi *m, uint32_t width, uint32_t height) {
    if (!m) return 0;
    GrvRenderParams geom{};
    geom.width = width;
    geom.height = height;
    geom.tile_world = (uint32_t)m->G;
    size_t px = 0;
    for (int r = (m->self_exchange ? 0 : 1); r < m->G; ++r) {
        geom.tile_rank = (uint32_t)r;
        px += grv_frame_ray_count(&geom);
    }
    if (m->G == 1 && !m->self_exchange) px = 0;
    return px * (m->format == GRV_EXCHANGE_RGBA16F ? 8u : 16u);
}

int grv_multi_rank_frame_stats(grv_multi *m, int rank, GrvFrameStats *out) {
    if (!m || !out || rank < 0 || rank >= m->G) return GRV_ERR_INVALID;
    int rc = grv_multi_synchronize(m);
    if (rc != GRV_OK) return rc;
    rc = grv_frame_stats(m->eng[rank], m->rank[rank].s[0], out);
    if (rc != GRV_OK) return mfail(m, rc, "rank %d: %s", rank, grv_last_error(m->eng[rank]));
    return GRV_OK;
}
grv_engine *grv_multi_engine(grv_multi *m, int rank) {
    return (m && rank >= 0 && rank < m->G) ? m->eng[rank] : nullptr;
}

int grv_multi_update_params(grv_multi *m, double mass, double spin) {
    if (!m) return GRV_ERR_INVALID;
    for (auto *e : m->eng) {
        const int rc = grv_update_params(e, mass, spin);
        if (rc != GRV_OK) return rc;
    }
    return GRV_OK;
}

int grv_render_frame_multi_device(grv_multi *m, const GrvCamera *cam, const GrvRenderParams *p, float *d_rgba,
                                  void *root_stream) {
    if (!m) return GRV_ERR_INVALID;
    if (!cam || !p) return mfail(m, GRV_ERR_INVALID, "null argument");
    if (p->tile_world > 1) return mfail(m, GRV_ERR_INVALID, "the multi-GPU entry deals the tiles itself: tile_world must be 0 or 1");
    const GrvRenderParams base = *p;
    const GrvCamera camera = *cam;
    const int G = m->G;
    return run_frame(m, p->width, p->height, d_rgba, static_cast<hipStream_t>(root_stream),
                     [&](int r, grv_engine *e, float *target, hipStream_t s) -> int {
                         GrvRenderParams rp = base;
                         if (G > 1 || m->self_exchange) {
                             rp.tile_world = (uint32_t)G;
                             rp.tile_rank = (uint32_t)r;
                         }
                         GrvFrameBuffers fb{};
                         fb.rgba = target;
                         return grv_render_frame_device(e, &camera, &rp, &fb, s);
                     });
}

int grv_render_frame_wgsl_multi_device(grv_multi *m, const GrvWgslParams *p, float *d_rgba, void *root_stream) {
    if (!m) return GRV_ERR_INVALID;
    if (!p) return mfail(m, GRV_ERR_INVALID, "null argument");
    if (p->tile_world > 1) return mfail(m, GRV_ERR_INVALID, "the multi-GPU entry deals the tiles itself: tile_world must be 0 or 1");
    const GrvWgslParams base = *p;
    const int G = m->G;
    return run_frame(m, p->width, p->height, d_rgba, static_cast<hipStream_t>(root_stream),
                     [&](int r, grv_engine *e, float *target, hipStream_t s) -> int {
                         GrvWgslParams rp = base;
                         if (G > 1 || m->self_exchange) {
                             rp.tile_world = (uint32_t)G;
                             rp.tile_rank = (uint32_t)r;
                         }
                         return grv_render_frame_wgsl(e, &rp, target, nullptr, nullptr, s);
                     });
}

int grv_multi_synchronize(grv_multi *m) {
    if (!m) return GRV_ERR_INVALID;
    for (int r = 0; r < m->G; ++r) {
        GRVM_HIP(m, hipSetDevice(m->dev[r]));
        for (int b = 0; b < 2; ++b) GRVM_HIP(m, hipStreamSynchronize(m->rank[r].s[b]));
    }
    GRVM_HIP(m, hipSetDevice(m->dev[0]));
    for (int b = 0; b < 2; ++b) GRVM_HIP(m, hipStreamSynchronize(m->rs[b]));
    return GRV_OK;
}

int grv_multi_stats_accumulate(grv_multi *m, int enable) {
    if (!m) return GRV_ERR_INVALID;
    for (auto *e : m->eng) grv_stats_accumulate(e, enable);
    return GRV_OK;
}

int grv_multi_frame_stats_reset(grv_multi *m) {
    if (!m) return GRV_ERR_INVALID;
    int rc = grv_multi_synchronize(m);
    if (rc != GRV_OK) return rc;
    for (int r = 0; r < m->G; ++r) {
        rc = grv_frame_stats_reset(m->eng[r], m->rank[r].s[0]);
        if (rc != GRV_OK) return mfail(m, rc, "rank %d: %s", r, grv_last_error(m->eng[r]));
    }
    return grv_multi_synchronize(m);
}

// Sums over the ranks (max for max_drift and for the event times: the ranks run side by side).
int grv_multi_frame_stats(grv_multi *m, GrvFrameStats *out) {
    if (!m || !out) return GRV_ERR_INVALID;
    int rc = grv_multi_synchronize(m);
    if (rc != GRV_OK) return rc;
    std::memset(out, 0, sizeof *out);
    for (int r = 0; r < m->G; ++r) {
        GrvFrameStats st;
        rc = grv_frame_stats(m->eng[r], m->rank[r].s[0], &st);
        if (rc != GRV_OK) return mfail(m, rc, "rank %d: %s", r, grv_last_error(m->eng[r]));
        out->rays += st.rays;
        out->accepted_steps += st.accepted_steps;
        out->rkf_tries += st.rkf_tries;
        for (int k = 0; k < 5; ++k) out->term_count[k] += st.term_count[k];
        out->crossings += st.crossings;
        out->max_drift = st.max_drift > out->max_drift ? st.max_drift : out->max_drift;
        out->launches += st.launches;
        out->init_ms = std::fmax(out->init_ms, st.init_ms);
        out->integrate_ms = std::fmax(out->integrate_ms, st.integrate_ms);
        out->compact_ms = std::fmax(out->compact_ms, st.compact_ms);
        out->shade_ms = std::fmax(out->shade_ms, st.shade_ms);
        out->total_ms = std::fmax(out->total_ms, st.total_ms);
    }
    return GRV_OK;
}

int grv_render_frame_multi(grv_multi *m, const GrvCamera *cam, const GrvRenderParams *p, float *rgba_host,
                           GrvFrameStats *stats) {
    if (!m) return GRV_ERR_INVALID;
    if (!p || !rgba_host) return mfail(m, GRV_ERR_INVALID, "null argument");
    const size_t px = (size_t)p->width * p->height;
    GRVM_HIP(m, hipSetDevice(m->dev[0]));
    if (px > m->image_px) {
        if (m->image) (void)hipFree(m->image);
        m->image = nullptr;
        m->image_px = 0;
        GRVM_HIP(m, hipMalloc(reinterpret_cast<void **>(&m->image), px * 16u));
        m->image_px = px;
    }
    hipStream_t s = m->rs[0];
    int rc = grv_render_frame_multi_device(m, cam, p, m->image, s);
    if (rc != GRV_OK) return rc;
    GRVM_HIP(m, hipSetDevice(m->dev[0]));
    GRVM_HIP(m, hipMemcpyAsync(rgba_host, m->image, px * 16u, hipMemcpyDeviceToHost, s));
    GRVM_HIP(m, hipStreamSynchronize(s));
    if (stats) return grv_multi_frame_stats(m, stats);
    return GRV_OK;
}

} // extern "C"
