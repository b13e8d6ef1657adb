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

#include <functional>
#include <mutex>

#include "multi_core.hpp"

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

// The HIP / RCCL side of grvmulti::Core (multi_core.hpp lists what an Api provides).
struct HipApi {
    using Stream = hipStream_t;
    using Event = hipEvent_t;
    std::vector<ncclComm_t> comm;
    const char *error_text() const { return msg().c_str(); }
    static std::string &msg() {
        static thread_local std::string m;
        return m;
    }
    static int hip(hipError_t st) {
        if (st == hipSuccess) return GRV_OK;
        msg() = hipGetErrorString(st);
        (void)hipGetLastError();
        return st == hipErrorOutOfMemory ? GRV_ERR_OOM : GRV_ERR_HIP;
    }
    static int nccl(ncclResult_t st) {
        if (st == ncclSuccess) return GRV_OK;
        msg() = g_rccl.GetErrorString(st);
        return GRV_ERR_HIP;
    }
    int set_device(int d) { return hip(hipSetDevice(d)); }
    int device_synchronize() { return hip(hipDeviceSynchronize()); }
    int malloc(void **p, size_t bytes) { return hip(hipMalloc(p, bytes)); }
    void free(void *p) { (void)hipFree(p); }
    int stream_wait_event(Stream s, Event e) { return hip(hipStreamWaitEvent(s, e, 0)); }
    int event_record(Event e, Stream s) { return hip(hipEventRecord(e, s)); }
    int copy_to_rank0(void *dst, int dst_dev, const void *src, int src_dev, size_t bytes, Stream s) {
        return hip(dst_dev == src_dev ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s)
                                      : hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, s));
    }
    int copy_on_device(void *dst, const void *src, size_t bytes, Stream s) {
        return hip(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
    }
    int pack_half(const float *src, void *dst, size_t n_px, Stream s) { return hip(launch_pack_half(src, dst, n_px, s)); }
    int widen_half(const void *src, float *dst, size_t n_px, Stream s) { return hip(launch_widen_half(src, dst, n_px, s)); }
    int quantize(float *img, size_t n_px, Stream s) { return hip(launch_post_quantize(img, (uint32_t)n_px, s)); }
    static GrvRenderParams geom_of(uint32_t width, uint32_t height, int G, int r) {
        GrvRenderParams geom{};
        geom.width = width;
        geom.height = height;
        geom.tile_world = (uint32_t)G;
        geom.tile_rank = (uint32_t)r;
        return geom;
    }
    int unpack_tiles(uint32_t width, uint32_t height, int G, int r, const void *slot, float *image, bool half, Stream s) {
        FrameGeom FG;
        frame_geometry(geom_of(width, height, G, r), FG);
        return hip(half ? launch_unpack_tiles_half(FG, slot, image, s) : launch_unpack_tiles(FG, slot, image, 4u, s));
    }
    size_t share_pixels(uint32_t width, uint32_t height, int G, int r) {
        const GrvRenderParams geom = geom_of(width, height, G, r);
        return grv_frame_ray_count(&geom);
    }
    size_t slot_pixels(uint32_t width, uint32_t height, int G) {
        const size_t total = (size_t)tile_pitch(width, (uint32_t)G) * ((height + 63u) / 64u);
        return (total + (size_t)G - 1) / (size_t)G * 4096u;
    }
    int group_start() { return nccl(g_rccl.GroupStart()); }
    int group_end() { return nccl(g_rccl.GroupEnd()); }
    int send(const void *src, size_t n, bool half, int r, Stream s) {
        return nccl(g_rccl.Send(src, n, half ? ncclHalf : ncclFloat, 0, comm[r], s));
    }
    int recv(void *dst, size_t n, bool half, int from, Stream s) {
        return nccl(g_rccl.Recv(dst, n, half ? ncclHalf : ncclFloat, from, comm[0], s));
    }
};

} // namespace

static_assert(grvmulti::OK == GRV_OK && grvmulti::ERR_INVALID == GRV_ERR_INVALID && grvmulti::ERR_HIP == GRV_ERR_HIP, "status codes");
static_assert(grvmulti::TRANSPORT_RCCL == GRV_TRANSPORT_RCCL && grvmulti::TRANSPORT_PEER_COPY == GRV_TRANSPORT_PEER_COPY, "transports");
static_assert(grvmulti::FORMAT_RGBA32F == GRV_EXCHANGE_RGBA32F && grvmulti::FORMAT_RGBA16F == GRV_EXCHANGE_RGBA16F, "formats");
static_assert(grvmulti::FAULT_RENDER == GRV_FAULT_RENDER && grvmulti::FAULT_SEND == GRV_FAULT_SEND &&
              grvmulti::FAULT_PEER_COPY == GRV_FAULT_PEER_COPY, "fault kinds");

struct grv_multi {
    grvmulti::Core<HipApi> c; // rank threads, exchange buffers, the frame skeleton (multi_core.hpp)
    bool virtual_ranks = false;
    std::vector<grv_engine *> eng;
    float *image = nullptr; // host-pointer entry: assembled image on rank 0's device
    size_t image_px = 0;
};

namespace {

int mfail(grv_multi *m, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (m) m->c.err = buf;
    return code;
}

#define GRVM_HIP(m, call)                                                                    \
    do {                                                                                     \
        hipError_t _st = (call);                                                             \
        if (_st != hipSuccess)                                                               \
            return mfail((m), _st == hipErrorOutOfMemory ? GRV_ERR_OOM : GRV_ERR_HIP,         \
                         "%s failed: %s", #call, hipGetErrorString(_st));                    \
    } while (0)

int run_frame(grv_multi *m, uint32_t width, uint32_t height, float *d_rgba, hipStream_t caller,
              const std::function<int(int, grv_engine *, float *, hipStream_t)> &render) {
    return m->c.run_frame(
        width, height, d_rgba, caller, [&](int r, float *target, hipStream_t s) { return render(r, m->eng[r], target, s); },
        [&](int r) { return std::string(grv_last_error(m->eng[r])); });
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
    m->c.G = G;
    m->c.dev = devs;
    m->virtual_ranks = virt;
    m->c.transport = transport;
    m->eng.assign(G, nullptr);
    m->c.rank.resize(G);
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
            GRVC_HIP(hipStreamCreateWithFlags(&m->c.rank[r].s[b], hipStreamNonBlocking), "hipStreamCreate", r);
            GRVC_HIP(hipEventCreateWithFlags(&m->c.rank[r].arrived[b], hipEventDisableTiming), "hipEventCreate", r);
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
        GRVC_HIP(hipStreamCreateWithFlags(&m->c.rs[b], hipStreamNonBlocking), "hipStreamCreate (exchange)", 0);
        GRVC_HIP(hipEventCreateWithFlags(&m->c.unpacked[b], hipEventDisableTiming), "hipEventCreate (exchange)", 0);
    }
    GRVC_HIP(hipEventCreateWithFlags(&m->c.caller_ready, hipEventDisableTiming), "hipEventCreate (caller)", 0);
#undef GRVC_HIP
    if (transport == GRV_TRANSPORT_RCCL) {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        // never a silent fall back to peer copies: a handle that asked for RCCL (AUTO between real
        // devices included) and cannot have it is refused, with the reason
        if (!g_rccl.load())
            return bail(cfail(GRV_ERR_NO_DEVICE, "RCCL transport unavailable: %s", g_rccl.err.c_str()));
        m->c.api.comm.assign(G, nullptr);
        const ncclResult_t st = g_rccl.CommInitAll(m->c.api.comm.data(), G, devs.data());
        if (st != ncclSuccess) {
            m->c.api.comm.clear();
            return bail(cfail(GRV_ERR_HIP, "ncclCommInitAll over %d devices failed: %s", G, g_rccl.GetErrorString(st)));
        }
    }
    m->c.threads = new (std::nothrow) grvmulti::RankThreads(G);
    if (!m->c.threads) return bail(cfail(GRV_ERR_OOM, "multi-GPU handle: out of host memory (rank threads)"));
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
    delete m->c.threads; // joins the workers
    m->c.threads = nullptr;
    for (int r = 0; r < m->c.G; ++r) {
        if (hipSetDevice(m->c.dev[r]) != hipSuccess) continue;
        (void)hipDeviceSynchronize();
    }
    if (!m->c.api.comm.empty())
        for (auto c : m->c.api.comm)
            if (c) (void)g_rccl.CommDestroy(c);
    for (int r = 0; r < m->c.G; ++r) {
        (void)hipSetDevice(m->c.dev[r]);
        for (int b = 0; b < 2; ++b) {
            if (m->c.rank[r].send[b]) (void)hipFree(m->c.rank[r].send[b]);
            if (m->c.rank[r].send16[b]) (void)hipFree(m->c.rank[r].send16[b]);
            if (m->c.rank[r].s[b]) (void)hipStreamDestroy(m->c.rank[r].s[b]);
            if (m->c.rank[r].arrived[b]) (void)hipEventDestroy(m->c.rank[r].arrived[b]);
        }
        if (m->eng[r]) grv_engine_destroy(m->eng[r]);
    }
    (void)hipSetDevice(m->c.dev[0]);
    for (int b = 0; b < 2; ++b) {
        if (m->c.recv[b]) (void)hipFree(m->c.recv[b]);
        if (m->c.rs[b]) (void)hipStreamDestroy(m->c.rs[b]);
        if (m->c.unpacked[b]) (void)hipEventDestroy(m->c.unpacked[b]);
    }
    if (m->c.caller_ready) (void)hipEventDestroy(m->c.caller_ready);
    if (m->image) (void)hipFree(m->image);
    delete m;
}

const char *grv_multi_last_error(const grv_multi *m) { return m ? m->c.err.c_str() : "null handle"; }
int grv_multi_rank_count(const grv_multi *m) { return m ? m->c.G : 0; }
int grv_multi_rank_device(const grv_multi *m, int rank) {
    return (m && rank >= 0 && rank < m->c.G) ? m->c.dev[rank] : -1;
}
int grv_multi_transport(const grv_multi *m) { return m ? m->c.transport : -1; }

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
    rc = m->c.drop_buffers();
    if (rc != GRV_OK) return rc;
    m->c.self_exchange = enable != 0;
    return GRV_OK;
}

int grv_multi_test_inject_fault(grv_multi *m, int kind, int rank) {
    if (!m) return GRV_ERR_INVALID;
    if (!grv_test_hooks_unlocked())
        return mfail(m, GRV_ERR_INVALID, "grv_multi_test_inject_fault: verification hooks are locked (grv_test_hooks_unlock)");
    if (kind < GRV_FAULT_NONE || kind > GRV_FAULT_PEER_COPY || rank < 0 || rank >= m->c.G)
        return mfail(m, GRV_ERR_INVALID, "grv_multi_test_inject_fault: kind %d / rank %d out of range", kind, rank);
    m->c.fault_kind = kind;
    m->c.fault_rank = rank;
    return GRV_OK;
}

int grv_multi_set_exchange_format(grv_multi *m, int format) {
    if (!m) return GRV_ERR_INVALID;
    if (format != GRV_EXCHANGE_RGBA32F && format != GRV_EXCHANGE_RGBA16F)
        return mfail(m, GRV_ERR_INVALID, "unknown exchange format %d", format);
    if (format == m->c.format) return GRV_OK;
    int rc = grv_multi_synchronize(m);
    if (rc != GRV_OK) return rc;
    rc = m->c.drop_buffers(); // sized for the other pixel width
    if (rc != GRV_OK) return rc;
    m->c.format = format;
    return GRV_OK;
}
int grv_multi_exchange_format(const grv_multi *m) { return m ? m->c.format : -1; }
size_t grv_multi_exchange_bytes_per_frame(const grv_multi *m, uint32_t width, uint32_t height) {
    if (!m) return 0;
    GrvRenderParams geom{};
    geom.width = width;
    geom.height = height;
    geom.tile_world = (uint32_t)m->c.G;
    size_t px = 0;
    for (int r = (m->c.self_exchange ? 0 : 1); r < m->c.G; ++r) {
        geom.tile_rank = (uint32_t)r;
        px += grv_frame_ray_count(&geom);
    }
    if (m->c.G == 1 && !m->c.self_exchange) px = 0;
    return px * (m->c.format == GRV_EXCHANGE_RGBA16F ? 8u : 16u);
}

int grv_multi_rank_frame_stats(grv_multi *m, int rank, GrvFrameStats *out) {
    if (!m || !out || rank < 0 || rank >= m->c.G) return GRV_ERR_INVALID;
    int rc = grv_multi_synchronize(m);
    if (rc != GRV_OK) return rc;
    rc = grv_frame_stats(m->eng[rank], m->c.rank[rank].s[0], out);
    if (rc != GRV_OK) return mfail(m, rc, "rank %d: %s", rank, grv_last_error(m->eng[rank]));
    return GRV_OK;
}
grv_engine *grv_multi_engine(grv_multi *m, int rank) {
    return (m && rank >= 0 && rank < m->c.G) ? m->eng[rank] : nullptr;
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
    const int G = m->c.G;
    return run_frame(m, p->width, p->height, d_rgba, static_cast<hipStream_t>(root_stream),
                     [&](int r, grv_engine *e, float *target, hipStream_t s) -> int {
                         GrvRenderParams rp = base;
                         if (G > 1 || m->c.self_exchange) {
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
    const int G = m->c.G;
    return run_frame(m, p->width, p->height, d_rgba, static_cast<hipStream_t>(root_stream),
                     [&](int r, grv_engine *e, float *target, hipStream_t s) -> int {
                         GrvWgslParams rp = base;
                         if (G > 1 || m->c.self_exchange) {
                             rp.tile_world = (uint32_t)G;
                             rp.tile_rank = (uint32_t)r;
                         }
                         return grv_render_frame_wgsl(e, &rp, target, nullptr, nullptr, s);
                     });
}

int grv_multi_synchronize(grv_multi *m) {
    if (!m) return GRV_ERR_INVALID;
    for (int r = 0; r < m->c.G; ++r) {
        GRVM_HIP(m, hipSetDevice(m->c.dev[r]));
        for (int b = 0; b < 2; ++b) GRVM_HIP(m, hipStreamSynchronize(m->c.rank[r].s[b]));
    }
    GRVM_HIP(m, hipSetDevice(m->c.dev[0]));
    for (int b = 0; b < 2; ++b) GRVM_HIP(m, hipStreamSynchronize(m->c.rs[b]));
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
    for (int r = 0; r < m->c.G; ++r) {
        rc = grv_frame_stats_reset(m->eng[r], m->c.rank[r].s[0]);
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
    for (int r = 0; r < m->c.G; ++r) {
        GrvFrameStats st;
        rc = grv_frame_stats(m->eng[r], m->c.rank[r].s[0], &st);
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
    GRVM_HIP(m, hipSetDevice(m->c.dev[0]));
    if (px > m->image_px) {
        if (m->image) (void)hipFree(m->image);
        m->image = nullptr;
        m->image_px = 0;
        GRVM_HIP(m, hipMalloc(reinterpret_cast<void **>(&m->image), px * 16u));
        m->image_px = px;
    }
    hipStream_t s = m->c.rs[0];
    int rc = grv_render_frame_multi_device(m, cam, p, m->image, s);
    if (rc != GRV_OK) return rc;
    GRVM_HIP(m, hipSetDevice(m->c.dev[0]));
    GRVM_HIP(m, hipMemcpyAsync(rgba_host, m->image, px * 16u, hipMemcpyDeviceToHost, s));
    GRVM_HIP(m, hipStreamSynchronize(s));
    if (stats) return grv_multi_frame_stats(m, stats);
    return GRV_OK;
}

} // extern "C"
