// engine.hip -- host side of libgravitas_hip.so: the C ABI of include/gravitas_abi.h
// on top of the segment kernels.  One engine == one `PhysicsEngine`
// (physics-engine/gravitas-wasm/src/lib.rs:42-54) bound to one HIP device.
//
// No CPU compute path exists here: every integrate / render / LUT entry point
// launches HIP kernels and fails with a status code if the device is missing.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "control_plane.hpp"
#include "spacetime_viz.hpp"
#include "engine_types.hpp"

using namespace grvhip;

// ---------------------------------------------------------------------------
// closed forms (host scalars; gravitas-core/src/metric/{mod,kerr}.rs)
// ---------------------------------------------------------------------------
namespace {

double clamp_rs(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

// metric/mod.rs:75-84
double event_horizon(double m, double spin) {
    const double a = spin * m;
    const double disc = m * m - a * a;
    return disc < 0.0 ? m : m + std::sqrt(disc);
}
// metric/kerr.rs:100-123 (prograde)
double isco_prograde(double m, double a_star) {
    if (std::fabs(a_star) < 1e-6) return m * 6.0;
    const double a2 = a_star * a_star;
    const double z1 = 1.0 + std::pow(1.0 - a2, 1.0 / 3.0) *
                                (std::pow(1.0 + a_star, 1.0 / 3.0) + std::pow(1.0 - a_star, 1.0 / 3.0));
    const double z2 = std::sqrt(3.0 * a2 + z1 * z1);
    const double disc = (3.0 - z1) * (3.0 + z1 + 2.0 * z2);
    const double root = disc < 0.0 ? 0.0 : std::sqrt(disc);
    return m * (3.0 + z2 - root);
}
// metric/kerr.rs:91-94
double photon_sphere(double m, double a_star) {
    const double term = (2.0 / 3.0) * std::acos(-a_star);
    return 2.0 * m * (1.0 + std::cos(term));
}
// metric/kerr.rs:181-189 on covariant_bl g_tt (kerr.rs:241-254), theta = pi/2;
// gravitas-wasm/src/lib.rs:97-105
double dilation(double m, double a_star, double r) {
    const double a = a_star * m;
    const double theta = 1.57079632679489661923;
    const double cos_theta = std::cos(theta);
    const double sigma = r * r + a * a * (cos_theta * cos_theta);
    const double g_tt = -(1.0 - (2.0 * m * r) / sigma);
    const double td = g_tt >= 0.0 ? 0.0 : std::sqrt(-g_tt);
    return td <= 0.0 ? 100.0 : 1.0 / td;
}
// physics/redshift.rs:65-95
double g_factor(double r, double mass, double spin, double lambda) {
    const double a = spin * mass, r2 = r * r, a2 = a * a, m = mass;
    const double omega = std::sqrt(m) / (std::pow(r, 1.5) + a * std::sqrt(m));
    const double sigma = r2;
    const double g_tt = -(1.0 - 2.0 * m * r / sigma);
    const double g_tphi = -(2.0 * m * r * a) / sigma;
    const double g_phiphi = r2 + a2 + 2.0 * m * r * a2 / sigma;
    const double ut_denom = -g_tt - 2.0 * omega * g_tphi - omega * omega * g_phiphi;
    if (ut_denom <= 0.0) return 0.0;
    const double ut = 1.0 / std::sqrt(ut_denom);
    const double factor = 1.0 - lambda * omega;
    if (std::fabs(factor) < 1e-30) return 0.0;
    return 1.0 / (ut * factor);
}

constexpr size_t kSabFloats = 2048; // lib.rs:67
constexpr size_t kOffControl = 0, kOffCamera = 64, kOffPhysics = 128, kOffTelemetry = 256,
                 kOffLuts = 2048; // lib.rs:36-40

} // namespace

// ---------------------------------------------------------------------------
// engine object
// ---------------------------------------------------------------------------
struct grv_engine {
    int device = 0;
    double mass = 1.0;
    double spin = 0.0;   // as given (lib.rs:44-45)
    double spin_c = 0.0; // clamped copy held by the metrics (kerr.rs:48-63)
    int n_cu = 256;
    std::string err;

    // ray workspace (device)
    void *ws_mem = nullptr;
    size_t ws_slots = 0;
    RayWorkspace ws{};
    uint32_t *live[2] = {nullptr, nullptr};
    uint32_t *d_counters = nullptr; // [0],[1] live counts (ping-pong)
    FrameStatsDev *d_stats = nullptr;
    uint32_t *h_counters = nullptr; // pinned
    FrameStatsDev *h_stats = nullptr; // pinned

    // staging buffers for host-pointer entry points
    void *stage_mem = nullptr;
    size_t stage_bytes = 0;

    // cached spectrum LUT (device)
    float *d_lut = nullptr;
    uint32_t lut_w = 0, lut_h = 0;
    double lut_tmax = 0.0;

    // last-frame bookkeeping
    uint32_t last_launches = 0;
    float last_ms[5] = {0, 0, 0, 0, 0};
    hipEvent_t ev[8] = {};
    bool ev_ok = false;

    // renderer layer (grv_webgpu_render / grv_webgl_render): full-size RGBA f32 targets
    struct Targets {
        float *mem = nullptr; // [3][h][w][4]: scene / compute texture, history ping, history pong
        uint32_t w = 0, h = 0;
        uint32_t hist = 0;   // webgpu: currentHistoryIndex; webgl: currentWriteIndex
        uint32_t frames = 0; // frameCount
    } rt;
    void *post_mem = nullptr; // bloom render targets (bright, blur ping/pong)
    size_t post_bytes = 0;
    uint8_t *d_noise = nullptr; // [2][256*256] R planes: u_noiseTex, u_blueNoiseTex
    std::vector<float> disk_lut = std::vector<float>(512, 0.0f); // lut_buffer (lib.rs:50, 65-66)
    std::vector<float> sab;
    float *sab_ext = nullptr; // attach_sab (lib.rs:74)
    CameraFilter camera, last_good_camera;
    float *sab_block() { return sab_ext ? sab_ext : sab.data(); }
};

namespace {

int fail(grv_engine *e, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (e) e->err = buf;
    return code;
}

#define GRV_HIP(e, call)                                                                   \
    do {                                                                                   \
        hipError_t _st = (call);                                                           \
        if (_st != hipSuccess)                                                             \
            return fail((e), _st == hipErrorOutOfMemory ? GRV_ERR_OOM : GRV_ERR_HIP,        \
                        "%s failed: %s", #call, hipGetErrorString(_st));                   \
    } while (0)

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int ensure_workspace(grv_engine *e, size_t slots) {
    if (slots <= e->ws_slots && e->ws_mem) {
        e->ws.n = (uint32_t)slots;
        return GRV_OK;
    }
    GRV_HIP(e, hipSetDevice(e->device));
    if (e->ws_mem) {
        (void)hipFree(e->ws_mem);
        e->ws_mem = nullptr;
        e->ws_slots = 0;
    }
    // 10 f64 components + crossing records + 3 u32 + two live lists, each 256-B aligned
    const size_t cap = align_up(slots, 64);
    const size_t f64b = align_up(cap * sizeof(double), 256);
    const size_t u32b = align_up(cap * sizeof(uint32_t), 256);
    const size_t total = f64b * (10 + kMaxCrossRec) + u32b * 5;
    void *mem = nullptr;
    GRV_HIP(e, hipMalloc(&mem, total));
    char *p = static_cast<char *>(mem);
    auto take64 = [&](size_t n) {
        double *q = reinterpret_cast<double *>(p);
        p += f64b * n;
        return q;
    };
    auto take32 = [&]() {
        uint32_t *q = reinterpret_cast<uint32_t *>(p);
        p += u32b;
        return q;
    };
    RayWorkspace w{};
    w.t = take64(1);
    w.r = take64(1);
    w.th = take64(1);
    w.ph = take64(1);
    w.pr = take64(1);
    w.pth = take64(1);
    w.pt = take64(1);
    w.pph = take64(1);
    w.h = take64(1);
    w.drift = take64(1);
    w.rc = take64(kMaxCrossRec);
    w.steps = take32();
    w.tries = take32();
    w.flags = take32();
    e->live[0] = take32();
    e->live[1] = take32();
    // rc rows are addressed as rc[c * n + slot]: keep n == capacity for the row pitch
    e->ws_mem = mem;
    e->ws_slots = cap;
    e->ws = w;
    e->ws.n = (uint32_t)slots;
    return GRV_OK;
}

int ensure_stage(grv_engine *e, size_t bytes) {
    if (bytes <= e->stage_bytes) return GRV_OK;
    if (e->stage_mem) (void)hipFree(e->stage_mem);
    e->stage_mem = nullptr;
    e->stage_bytes = 0;
    GRV_HIP(e, hipMalloc(&e->stage_mem, bytes));
    e->stage_bytes = bytes;
    return GRV_OK;
}

SegmentParams make_segment_params(const grv_engine *e, const GrvOptions &o) {
    SegmentParams P{};
    const double spin = (o.metric_kind == GRV_METRIC_SCHWARZSCHILD) ? 0.0 : e->spin_c;
    P.M = e->mass;
    P.a = spin * e->mass; // kerr.rs:70-74
    P.a2 = P.a * P.a;
    P.horizon_limit = event_horizon(e->mass, spin) * 1.001; // geodesic/mod.rs:258
    P.escape_radius = o.escape_radius;
    P.tolerance = o.tolerance;
    P.inv_tolerance = 1.0 / o.tolerance;
    P.step_size = o.step_size;
    P.max_steps = o.max_steps > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)o.max_steps;
    P.renorm_interval =
        o.renormalize_interval > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)o.renormalize_interval;
    P.max_tries = 0;
    P.shading = 0;
    P.disk_inner = 0.0;
    P.disk_outer = 0.0;
    P.max_crossings = 0xFFFFFFFFu;
    return P;
}

bool options_valid(const GrvOptions &o) {
    if (o.method < GRV_METHOD_RKF45 || o.method > GRV_METHOD_SYMPLECTIC) return false;
    if (o.metric_kind < GRV_METRIC_KERR_BL || o.metric_kind > GRV_METRIC_SCHWARZSCHILD) return false;
    if (o.arith != GRV_ARITH_STRICT && o.arith != GRV_ARITH_FAST) return false;
    if (o.method == GRV_METHOD_RKF45 && !(o.tolerance > 0.0)) return false;
    return true;
}

hipError_t launch_segment(int arith, int kind, int method, const RayWorkspace &ws,
                          const SegmentParams &P, const uint32_t *live_in, uint32_t n_live,
                          uint32_t *live_out, uint32_t *cnt, hipStream_t s) {
    return arith == GRV_ARITH_FAST
               ? launch_segment_fast(kind, method, ws, P, live_in, n_live, live_out, cnt, s)
               : launch_segment_strict(kind, method, ws, P, live_in, n_live, live_out, cnt, s);
}

// Runs segments until no ray is live.  The workspace must have been initialised;
// the first launch walks every slot (identity live list), later launches walk the
// compacted list the previous one appended.
int run_segments(grv_engine *e, const GrvOptions &o, SegmentParams P, uint32_t seg_tries,
                 hipStream_t s, bool profile) {
    if (seg_tries == 0) seg_tries = 4096;
    P.max_tries = seg_tries;
    uint32_t n_live = e->ws.n;
    const uint32_t *live_in = nullptr;
    int cur = 0;
    e->last_launches = 0;
    // every live ray completes a step within <= 9 tries (<= 7 shrinks by >= 10x from
    // |h| <= 10 down to the forced 1e-5 step), so this bound is never reached.
    const uint64_t hard_cap = ((uint64_t)P.max_steps * 9ull) / seg_tries + 4ull;
    float integ_ms = 0.f;
    while (n_live > 0) {
        if (e->last_launches > hard_cap)
            return fail(e, GRV_ERR_HIP, "segment loop exceeded its bound (%u live)", n_live);
        const int nxt = cur ^ 1;
        GRV_HIP(e, hipMemsetAsync(e->d_counters + nxt, 0, sizeof(uint32_t), s));
        if (profile) GRV_HIP(e, hipEventRecord(e->ev[2], s));
        GRV_HIP(e, launch_segment(o.arith, o.metric_kind, o.method, e->ws, P, live_in, n_live,
                                  e->live[nxt], e->d_counters + nxt, s));
        if (profile) GRV_HIP(e, hipEventRecord(e->ev[3], s));
        GRV_HIP(e, hipMemcpyAsync(e->h_counters + nxt, e->d_counters + nxt, sizeof(uint32_t),
                                  hipMemcpyDeviceToHost, s));
        GRV_HIP(e, hipStreamSynchronize(s));
        if (profile) {
            float ms = 0.f;
            GRV_HIP(e, hipEventElapsedTime(&ms, e->ev[2], e->ev[3]));
            integ_ms += ms;
        }
        n_live = e->h_counters[nxt];
        live_in = e->live[nxt];
        cur = nxt;
        e->last_launches++;
    }
    e->last_ms[1] = integ_ms;
    return GRV_OK;
}

int ensure_lut(grv_engine *e, uint32_t w, uint32_t h, double tmax, hipStream_t s) {
    if (e->d_lut && e->lut_w == w && e->lut_h == h && e->lut_tmax == tmax) return GRV_OK;
    if (w == 0 || h == 0 || (uint64_t)w * h > (1ull << 26)) return fail(e, GRV_ERR_INVALID, "bad LUT shape");
    if (e->d_lut) (void)hipFree(e->d_lut);
    e->d_lut = nullptr;
    GRV_HIP(e, hipMalloc(reinterpret_cast<void **>(&e->d_lut), (size_t)w * h * 4 * sizeof(float)));
    GRV_HIP(e, launch_spectrum_lut(e->d_lut, w, h, tmax, s));
    e->lut_w = w;
    e->lut_h = h;
    e->lut_tmax = tmax;
    return GRV_OK;
}

void frame_geometry(const GrvRenderParams &p, FrameGeom &G) {
    G.width = p.width;
    G.height = p.height;
    G.tiles_x = (p.width + 63u) / 64u;
    G.tiles_y = (p.height + 63u) / 64u;
    G.tile_world = p.tile_world == 0 ? 1u : p.tile_world;
    G.tile_rank = p.tile_world == 0 ? 0u : p.tile_rank;
    const uint32_t total = G.tiles_x * G.tiles_y;
    G.n_tiles_local = (total > G.tile_rank) ? (total - G.tile_rank + G.tile_world - 1u) / G.tile_world : 0u;
}

void stats_to_abi(const grv_engine *e, const FrameStatsDev &d, GrvFrameStats *out) {
    std::memset(out, 0, sizeof *out);
    out->rays = d.rays;
    out->accepted_steps = d.accepted_steps;
    out->rkf_tries = d.rkf_tries;
    for (int k = 0; k < 5; ++k) out->term_count[k] = d.term_count[k];
    out->crossings = d.crossings;
    double md;
    std::memcpy(&md, &d.max_drift_bits, sizeof md);
    out->max_drift = md;
    out->launches = e->last_launches;
    out->init_ms = e->last_ms[0];
    out->integrate_ms = e->last_ms[1];
    out->compact_ms = e->last_ms[2];
    out->shade_ms = e->last_ms[3];
    out->total_ms = e->last_ms[4];
}

} // namespace

namespace {
template <typename Launch>
int run_shader_frame(grv_engine *e, uint32_t width, uint32_t height, uint32_t tw, uint32_t tr,
                     uint64_t *total_steps, hipStream_t s, Launch &&launch) {
    if (width == 0 || height == 0) return fail(e, GRV_ERR_INVALID, "empty frame");
    if (tw > 1 && tr >= tw) return fail(e, GRV_ERR_INVALID, "tile_rank >= tile_world");
    GRV_HIP(e, hipSetDevice(e->device));
    GrvRenderParams q{};
    q.width = width;
    q.height = height;
    q.tile_world = tw;
    q.tile_rank = tr;
    FrameGeom G;
    frame_geometry(q, G);
    const size_t slots = (size_t)G.n_tiles_local * 4096u;
    GRV_HIP(e, hipMemsetAsync(e->d_stats, 0, sizeof(FrameStatsDev), s));
    if (slots > 0x7FFFFFFFull) return fail(e, GRV_ERR_INVALID, "frame too large for one rank");
    GRV_HIP(e, launch(G, (uint32_t)slots, &e->d_stats->accepted_steps));
    if (total_steps) {
        GRV_HIP(e, hipMemcpyAsync(e->h_stats, e->d_stats, sizeof(FrameStatsDev), hipMemcpyDeviceToHost, s));
        GRV_HIP(e, hipStreamSynchronize(s));
        *total_steps = e->h_stats->accepted_steps;
    }
    return GRV_OK;
}
} // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
// ---- helpers of the spacetime read-outs (spacetime_viz.hip) ----
namespace {
VizHole viz_hole(const grv_engine *e) { return VizHole{e->mass, e->spin, e->spin_c * e->mass}; }

// run one of the grid kernels into the staging buffer and copy n_floats back
template <typename Launch>
int viz_grid(grv_engine *e, size_t n_a, size_t n_b, float *out, Launch &&launch) {
    if (!e) return GRV_ERR_INVALID;
    if (n_a == 0 || n_b == 0) return GRV_OK; // empty loops upstream
    if (!out) return fail(e, GRV_ERR_INVALID, "null output");
    if (n_a > 0xFFFFu * 16u || n_b > 0xFFFFu * 16u || n_a * n_b > (size_t)1 << 28)
        return fail(e, GRV_ERR_INVALID, "grid %zu x %zu too large", n_a, n_b);
    GRV_HIP(e, hipSetDevice(e->device));
    const size_t bytes = n_a * n_b * 3 * sizeof(float);
    int rc = ensure_stage(e, bytes);
    if (rc != GRV_OK) return rc;
    float *d_out = static_cast<float *>(e->stage_mem);
    GRV_HIP(e, launch(d_out));
    GRV_HIP(e, hipDeviceSynchronize());
    GRV_HIP(e, hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost));
    return GRV_OK;
}
} // namespace

extern "C" {

int grv_abi_version(void) { return GRV_ABI_VERSION; }

int grv_engine_create(double mass, double spin, int device, grv_engine **out) {
    if (!out) return GRV_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    hipError_t st = hipGetDeviceCount(&count);
    if (st != hipSuccess || count <= 0 || device < 0 || device >= count) return GRV_ERR_NO_DEVICE;
    grv_engine *e = new (std::nothrow) grv_engine();
    if (!e) return GRV_ERR_OOM;
    e->device = device;
    e->mass = mass;
    e->spin = spin;
    e->spin_c = clamp_rs(spin, -1.0, 1.0);
    e->sab.assign(kSabFloats, 0.0f);
    auto bail = [&](int code) {
        grv_engine_destroy(e);
        return code;
    };
    if (hipSetDevice(device) != hipSuccess) return bail(GRV_ERR_NO_DEVICE);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) e->n_cu = prop.multiProcessorCount;
    if (hipMalloc(reinterpret_cast<void **>(&e->d_counters), 4 * sizeof(uint32_t)) != hipSuccess) return bail(GRV_ERR_OOM);
    if (hipMalloc(reinterpret_cast<void **>(&e->d_stats), sizeof(FrameStatsDev)) != hipSuccess) return bail(GRV_ERR_OOM);
    if (hipHostMalloc(reinterpret_cast<void **>(&e->h_counters), 4 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) return bail(GRV_ERR_OOM);
    if (hipHostMalloc(reinterpret_cast<void **>(&e->h_stats), sizeof(FrameStatsDev), hipHostMallocDefault) != hipSuccess) return bail(GRV_ERR_OOM);
    std::memset(e->h_stats, 0, sizeof(FrameStatsDev));
    e->ev_ok = true;
    for (auto &ev : e->ev)
        if (hipEventCreate(&ev) != hipSuccess) e->ev_ok = false;
    *out = e;
    return GRV_OK;
}

void grv_engine_destroy(grv_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->ws_mem) (void)hipFree(e->ws_mem);
    if (e->stage_mem) (void)hipFree(e->stage_mem);
    if (e->d_lut) (void)hipFree(e->d_lut);
    if (e->d_noise) (void)hipFree(e->d_noise);
    if (e->post_mem) (void)hipFree(e->post_mem);
    if (e->rt.mem) (void)hipFree(e->rt.mem);
    if (e->d_counters) (void)hipFree(e->d_counters);
    if (e->d_stats) (void)hipFree(e->d_stats);
    if (e->h_counters) (void)hipHostFree(e->h_counters);
    if (e->h_stats) (void)hipHostFree(e->h_stats);
    if (e->ev_ok)
        for (auto &ev : e->ev) (void)hipEventDestroy(ev);
    delete e;
}

const char *grv_last_error(const grv_engine *e) { return e ? e->err.c_str() : "null engine"; }

int grv_update_params(grv_engine *e, double mass, double spin) {
    if (!e) return GRV_ERR_INVALID;
    e->mass = mass;
    e->spin = spin;
    e->spin_c = clamp_rs(spin, -1.0, 1.0);
    return GRV_OK;
}

double grv_compute_horizon(const grv_engine *e) { return event_horizon(e->mass, e->spin_c); }
double grv_compute_isco(const grv_engine *e) { return isco_prograde(e->mass, e->spin_c); }
double grv_compute_photon_sphere(const grv_engine *e) { return photon_sphere(e->mass, e->spin_c); }
double grv_compute_dilation(const grv_engine *e, double r) { return dilation(e->mass, e->spin_c, r); }
double grv_compute_g_factor(const grv_engine *e, double r, double lambda) {
    return g_factor(r, e->mass, e->spin, lambda); // lib.rs:203-205 passes self.spin unclamped
}

void grv_options_default(GrvOptions *o) {
    if (!o) return;
    std::memset(o, 0, sizeof *o);
    o->method = GRV_METHOD_RKF45;
    o->metric_kind = GRV_METRIC_KERR_KS;
    o->tolerance = 1e-8;
    o->initial_step = 0.01;
    o->max_steps = 10000;
    o->escape_radius = 1000.0;
    o->renormalize_interval = 10;
    o->step_size = 0.0;
    o->arith = GRV_ARITH_STRICT;
}

int grv_integrate_batch_device(grv_engine *e, size_t n, const double *d_states,
                               const GrvOptions *opt, double *d_out_states, uint32_t *d_steps,
                               uint8_t *d_termination, double *d_drift, void *stream) {
    if (!e) return GRV_ERR_INVALID;
    if (!opt || !options_valid(*opt)) return fail(e, GRV_ERR_INVALID, "invalid GrvOptions");
    if (n == 0) return GRV_OK;
    if (!d_states) return fail(e, GRV_ERR_INVALID, "null states");
    if (n > 0x7FFFFFFFull) return fail(e, GRV_ERR_INVALID, "batch too large");
    hipStream_t s = static_cast<hipStream_t>(stream);
    GRV_HIP(e, hipSetDevice(e->device));
    int rc = ensure_workspace(e, n);
    if (rc != GRV_OK) return rc;
    SegmentParams P = make_segment_params(e, *opt);
    GRV_HIP(e, hipMemsetAsync(e->d_counters, 0, 4 * sizeof(uint32_t), s));
    GRV_HIP(e, hipMemsetAsync(e->d_stats, 0, sizeof(FrameStatsDev), s));
    GRV_HIP(e, launch_init_states(opt->metric_kind, e->ws, P, d_states, opt->initial_step,
                                  opt->method == GRV_METHOD_RKF45, s));
    // independent rays diverge freely in a batch: compact every 64 tries unless told otherwise
    rc = run_segments(e, *opt, P, opt->segment_tries > 0 ? (uint32_t)opt->segment_tries : 64u, s, false);
    if (rc != GRV_OK) return rc;
    GRV_HIP(e, launch_finalize_batch(e->ws, d_out_states, d_steps, d_termination, d_drift,
                                     e->d_stats, s));
    return GRV_OK;
}

int grv_integrate_batch(grv_engine *e, size_t n, const double *states, const GrvOptions *opt,
                        double *out_states, uint32_t *steps, uint8_t *termination, double *drift) {
    if (!e) return GRV_ERR_INVALID;
    if (n == 0) return GRV_OK;
    if (!states || !out_states) return fail(e, GRV_ERR_INVALID, "null states");
    GRV_HIP(e, hipSetDevice(e->device));
    const size_t sb = align_up(n * 64, 256), ub = align_up(n * 4, 256), bb = align_up(n, 256),
                 db = align_up(n * 8, 256);
    int rc = ensure_stage(e, 2 * sb + ub + bb + db);
    if (rc != GRV_OK) return rc;
    char *p = static_cast<char *>(e->stage_mem);
    double *d_in = reinterpret_cast<double *>(p);
    double *d_out = reinterpret_cast<double *>(p + sb);
    uint32_t *d_steps = reinterpret_cast<uint32_t *>(p + 2 * sb);
    uint8_t *d_term = reinterpret_cast<uint8_t *>(p + 2 * sb + ub);
    double *d_drift = reinterpret_cast<double *>(p + 2 * sb + ub + bb);
    GRV_HIP(e, hipMemcpy(d_in, states, n * 64, hipMemcpyHostToDevice));
    rc = grv_integrate_batch_device(e, n, d_in, opt, d_out, d_steps, d_term, d_drift, nullptr);
    if (rc != GRV_OK) return rc;
    GRV_HIP(e, hipDeviceSynchronize());
    GRV_HIP(e, hipMemcpy(out_states, d_out, n * 64, hipMemcpyDeviceToHost));
    if (steps) GRV_HIP(e, hipMemcpy(steps, d_steps, n * 4, hipMemcpyDeviceToHost));
    if (termination) GRV_HIP(e, hipMemcpy(termination, d_term, n, hipMemcpyDeviceToHost));
    if (drift) GRV_HIP(e, hipMemcpy(drift, d_drift, n * 8, hipMemcpyDeviceToHost));
    return GRV_OK;
}

size_t grv_integrate_ray_relativistic(grv_engine *e, const double *initial_state, size_t n,
                                      size_t steps, double tolerance, int use_kerr_schild,
                                      double *out) {
    if (!e || !initial_state || !out) return 0;
    if (n < 8) { // lib.rs:429-431
        for (size_t i = 0; i < n; ++i) out[i] = initial_state[i];
        return n;
    }
    GrvOptions o; // lib.rs:444-452
    grv_options_default(&o);
    o.method = GRV_METHOD_RKF45;
    o.metric_kind = use_kerr_schild ? GRV_METRIC_KERR_KS : GRV_METRIC_KERR_BL;
    o.tolerance = tolerance;
    o.initial_step = 0.01;
    o.max_steps = steps;
    o.escape_radius = 1000.0;
    o.renormalize_interval = 10;
    o.arith = GRV_ARITH_STRICT;
    double res[8];
    if (grv_integrate_batch(e, 1, initial_state, &o, res, nullptr, nullptr, nullptr) != GRV_OK) {
        // no Result in the reference FFI: hand back NaNs so the caller's finite-guard
        // (src/engine/physics-bridge.ts:174-180) trips; the error text stays on the handle
        for (int i = 0; i < 8; ++i) out[i] = std::nan("");
        return 8;
    }
    std::memcpy(out, res, sizeof res);
    return 8;
}

size_t grv_frame_ray_count(const GrvRenderParams *p) {
    if (!p) return 0;
    FrameGeom G;
    frame_geometry(*p, G);
    if (G.tile_world <= 1) return (size_t)p->width * p->height;
    return (size_t)G.n_tiles_local * 4096u;
}

void grv_render_params_default(uint32_t width, uint32_t height, GrvRenderParams *p) {
    if (!p) return;
    std::memset(p, 0, sizeof *p);
    p->width = width;
    p->height = height;
    grv_options_default(&p->opt);
    p->opt.max_steps = 2048;
    p->shading = 1;
    p->precision = 0;
    p->disk_inner = 0.0;
    p->disk_outer = 30.0;
    p->disk_temp = 9500.0;
    p->disk_opacity = 0.6;
    p->exposure = 1.0;
    p->lut_width = 512;
    p->lut_height = 64;
    p->lut_max_temp = 1e5;
    p->tile_world = 1;
    p->tile_rank = 0;
    p->segment_tries = 0;
    p->profile = 0;
}

int grv_render_frame_device(grv_engine *e, const GrvCamera *cam, const GrvRenderParams *p,
                            const GrvFrameBuffers *out, void *stream) {
    if (!e) return GRV_ERR_INVALID;
    if (!cam || !p || !out) return fail(e, GRV_ERR_INVALID, "null argument");
    if (!options_valid(p->opt)) return fail(e, GRV_ERR_INVALID, "invalid GrvOptions");
    if (p->width == 0 || p->height == 0) return fail(e, GRV_ERR_INVALID, "empty frame");
    if (p->precision != 0) return fail(e, GRV_ERR_INVALID, "f32 frame kernels not built yet");
    if (p->tile_world > 1 && p->tile_rank >= p->tile_world) return fail(e, GRV_ERR_INVALID, "tile_rank >= tile_world");
    hipStream_t s = static_cast<hipStream_t>(stream);
    GRV_HIP(e, hipSetDevice(e->device));
    const bool profile = p->profile != 0 && e->ev_ok;

    FrameGeom G;
    frame_geometry(*p, G);
    const size_t slots = (size_t)G.n_tiles_local * 4096u;
    for (float &m : e->last_ms) m = 0.f;
    e->last_launches = 0;
    GRV_HIP(e, hipMemsetAsync(e->d_counters, 0, 4 * sizeof(uint32_t), s));
    GRV_HIP(e, hipMemsetAsync(e->d_stats, 0, sizeof(FrameStatsDev), s));
    if (slots == 0) return GRV_OK;
    if (slots > 0x7FFFFFFFull) return fail(e, GRV_ERR_INVALID, "frame too large for one rank");
    int rc = ensure_workspace(e, slots);
    if (rc != GRV_OK) return rc;

    SegmentParams P = make_segment_params(e, p->opt);
    const double spin = (p->opt.metric_kind == GRV_METRIC_SCHWARZSCHILD) ? 0.0 : e->spin_c;
    const double disk_inner = p->disk_inner > 0.0 ? p->disk_inner : isco_prograde(e->mass, e->spin_c);
    if (p->shading) {
        // smallest crossing count whose accumulated alpha exceeds 0.99 (alpha += opacity)
        uint32_t nmax = 0;
        double alpha = 0.0;
        while (nmax < 15) {
            alpha += p->disk_opacity;
            ++nmax;
            if (alpha > 0.99) break;
        }
        if (!(alpha > 0.99)) nmax = 0xFFFFFFFFu;
        if (nmax != 0xFFFFFFFFu && nmax > (uint32_t)kMaxCrossRec)
            return fail(e, GRV_ERR_INVALID, "disk_opacity too small: more than %d crossings to record", kMaxCrossRec);
        if (nmax == 0xFFFFFFFFu)
            return fail(e, GRV_ERR_INVALID, "disk_opacity must reach alpha > 0.99 within %d crossings", kMaxCrossRec);
        P.shading = 1;
        P.disk_inner = disk_inner;
        P.disk_outer = p->disk_outer;
        P.max_crossings = nmax;
        rc = ensure_lut(e, p->lut_width, p->lut_height, p->lut_max_temp, s);
        if (rc != GRV_OK) return rc;
    }

    CameraDev cd;
    std::memcpy(cd.pos, cam->position, sizeof cd.pos);
    std::memcpy(cd.inv_view, cam->inv_view, sizeof cd.inv_view);
    std::memcpy(cd.inv_proj, cam->inv_proj, sizeof cd.inv_proj);
    std::memcpy(cd.off, cam->pixel_offset, sizeof cd.off);

    if (profile) GRV_HIP(e, hipEventRecord(e->ev[0], s));
    GRV_HIP(e, launch_init_pixels(p->opt.metric_kind, e->ws, P, G, cd, p->opt.initial_step,
                                  p->opt.method == GRV_METHOD_RKF45, s));
    if (profile) GRV_HIP(e, hipEventRecord(e->ev[1], s));
    // neighbouring pixels take near-identical step counts (8x8-pixel waves run at >99 %
    // lane efficiency at 4K), so the frame default is one long segment; segment_tries
    // selects the compacting wavefront form
    rc = run_segments(e, p->opt, P, p->segment_tries, s, profile);
    if (rc != GRV_OK) return rc;

    ShadeParams S{};
    S.M = e->mass;
    S.spin = spin;
    S.disk_inner = disk_inner;
    S.disk_temp = p->disk_temp;
    S.disk_opacity = p->disk_opacity;
    S.exposure = p->exposure;
    S.lut_w = p->lut_width;
    S.lut_h = p->lut_height;
    S.lut_max_temp = p->lut_max_temp;
    if (p->shading) {
        // stage a band of g rows around g = 1 in LDS (128 KiB budget); the rest is
        // served from L2/HBM by the same lookup
        const uint32_t fit = (uint32_t)((128u * 1024u) / ((size_t)S.lut_w * 16u));
        S.lds_rows = fit < S.lut_h ? fit : S.lut_h;
        const double row_g1 = (1.0 - 0.05) / (5.0 - 0.05) * (double)(S.lut_h > 1 ? S.lut_h - 1 : 1);
        int r0 = (int)row_g1 - (int)(S.lds_rows * 2 / 3);
        if (r0 < 0) r0 = 0;
        if ((uint32_t)r0 + S.lds_rows > S.lut_h) r0 = (int)(S.lut_h - S.lds_rows);
        S.lds_row0 = (uint32_t)r0;
    }
    if (profile) GRV_HIP(e, hipEventRecord(e->ev[4], s));
    GRV_HIP(e, launch_finalize_frame(e->ws, G, S, p->shading, e->d_lut, out->rgba,
                                     out->final_state, out->steps, out->termination, out->drift,
                                     e->d_stats, e->n_cu, s));
    if (profile) {
        GRV_HIP(e, hipEventRecord(e->ev[5], s));
        GRV_HIP(e, hipEventSynchronize(e->ev[5]));
        GRV_HIP(e, hipEventElapsedTime(&e->last_ms[0], e->ev[0], e->ev[1]));
        GRV_HIP(e, hipEventElapsedTime(&e->last_ms[3], e->ev[4], e->ev[5]));
        GRV_HIP(e, hipEventElapsedTime(&e->last_ms[4], e->ev[0], e->ev[5]));
    }
    return GRV_OK;
}

int grv_frame_stats(grv_engine *e, void *stream, GrvFrameStats *stats) {
    if (!e || !stats) return GRV_ERR_INVALID;
    hipStream_t s = static_cast<hipStream_t>(stream);
    GRV_HIP(e, hipSetDevice(e->device));
    GRV_HIP(e, hipMemcpyAsync(e->h_stats, e->d_stats, sizeof(FrameStatsDev), hipMemcpyDeviceToHost, s));
    GRV_HIP(e, hipStreamSynchronize(s));
    stats_to_abi(e, *e->h_stats, stats);
    return GRV_OK;
}

int grv_render_frame(grv_engine *e, const GrvCamera *cam, const GrvRenderParams *p,
                     float *rgba_host, GrvFrameStats *stats) {
    if (!e) return GRV_ERR_INVALID;
    if (!p || !rgba_host) return fail(e, GRV_ERR_INVALID, "null argument");
    const size_t n = grv_frame_ray_count(p);
    GRV_HIP(e, hipSetDevice(e->device));
    int rc = ensure_stage(e, align_up(n * 16, 256));
    if (rc != GRV_OK) return rc;
    GrvFrameBuffers fb{};
    fb.rgba = static_cast<float *>(e->stage_mem);
    rc = grv_render_frame_device(e, cam, p, &fb, nullptr);
    if (rc != GRV_OK) return rc;
    GRV_HIP(e, hipDeviceSynchronize());
    GRV_HIP(e, hipMemcpy(rgba_host, fb.rgba, n * 16, hipMemcpyDeviceToHost));
    if (stats) return grv_frame_stats(e, nullptr, stats);
    return GRV_OK;
}

int grv_unpack_tiles(const GrvRenderParams *p, uint32_t rank, const void *packed, void *image,
                     size_t bpp) {
    if (!p || !packed || !image || bpp == 0) return GRV_ERR_INVALID;
    GrvRenderParams q = *p;
    q.tile_rank = rank;
    FrameGeom G;
    frame_geometry(q, G);
    const char *src = static_cast<const char *>(packed);
    char *dst = static_cast<char *>(image);
    for (uint32_t tl = 0; tl < G.n_tiles_local; ++tl) {
        const uint32_t tile = tl * G.tile_world + G.tile_rank;
        const uint32_t tx = tile % G.tiles_x, ty = tile / G.tiles_x;
        for (uint32_t py = 0; py < 64; ++py) {
            const uint32_t Y = ty * 64 + py;
            if (Y >= G.height) break;
            const uint32_t X0 = tx * 64;
            const uint32_t w = (X0 + 64 <= G.width) ? 64u : (G.width - X0);
            std::memcpy(dst + ((size_t)Y * G.width + X0) * bpp,
                        src + ((size_t)tl * 4096u + (size_t)py * 64u) * bpp, (size_t)w * bpp);
        }
    }
    return GRV_OK;
}

void grv_wgsl_params_default(uint32_t width, uint32_t height, const GrvCamera *cam, double mass,
                             double spin, GrvWgslParams *p) {
    if (!p) return;
    std::memset(p, 0, sizeof *p);
    p->width = width;
    p->height = height;
    if (cam) {
        for (int k = 0; k < 16; ++k) {
            p->inv_view[k] = (float)cam->inv_view[k];
            p->inv_proj[k] = (float)cam->inv_proj[k];
        }
        for (int k = 0; k < 3; ++k) p->position[k] = (float)cam->position[k];
    }
    p->mass = (float)mass;
    p->spin = (float)spin;
    p->max_steps = 150; // compute.wgsl.ts:13
    p->tile_world = 1;
}

void grv_glsl_params_default(uint32_t width, uint32_t height, double mass, double spin,
                             GrvGlslParams *p) {
    if (!p) return;
    std::memset(p, 0, sizeof *p);
    p->width = width;
    p->height = height;
    p->mass = (float)mass;
    p->spin = (float)(spin * mass);              // renderer.ts:326
    p->zoom = 30.0f * 2.0f;                       // simulation.config.ts:118-119, renderer.ts:327
    p->mouse[0] = 0.5f;
    p->mouse[1] = 97.0f / 180.0f;                 // simulation.config.ts:106-107
    p->disk_size = 50.0f;                         // simulation.config.ts:138-139
    p->disk_scale_height = 0.2f;                  // :147-148
    p->disk_density = 4.0f;                       // :167-168
    p->disk_temp = (float)(9500.0 * std::pow(mass, -0.25)); // renderer.ts:352-356
    p->lensing_strength = 1.0f;                   // renderer.ts:340
    p->time = 0.0f;
    p->turbulence = -1.0f;                        // sample the noise texture (disk.ts:55)
    p->max_ray_steps = 256;                       // simulation.config.ts:205-211 (ultra)
    p->tone_map = 0;
    p->tile_world = 1;
    p->features = GRV_GLSL_FEATURES_DEFAULT;
    p->quality = 1;
    p->cam_quat[3] = 1.0f;                        // renderer.ts:315-316
}

void grv_seeded_noise_rgba8(uint32_t seed, uint32_t size, uint8_t *rgba) {
    if (!rgba) return;
    // xorshift32 stream; byte = floor(u * 255), u in [0, 1), as createNoiseTexture forms it
    uint32_t x = seed ? seed : 0x9E3779B9u;
    const size_t n = (size_t)size * size * 4u;
    for (size_t i = 0; i < n; ++i) {
        x ^= x << 13;
        x ^= x >> 17;
        x ^= x << 5;
        rgba[i] = (uint8_t)std::floor((double)(x >> 8) / 16777216.0 * 255.0);
    }
}

int grv_set_glsl_noise(grv_engine *e, const uint8_t *noise_rgba, const uint8_t *blue_rgba) {
    if (!e) return GRV_ERR_INVALID;
    GRV_HIP(e, hipSetDevice(e->device));
    constexpr size_t kPlane = 256 * 256;
    if (!e->d_noise) {
        GRV_HIP(e, hipMalloc(reinterpret_cast<void **>(&e->d_noise), 2 * kPlane));
        GRV_HIP(e, hipMemset(e->d_noise, 0, 2 * kPlane));
    }
    std::vector<uint8_t> plane(kPlane);
    const uint8_t *src[2] = {noise_rgba, blue_rgba};
    for (int t = 0; t < 2; ++t) {
        if (!src[t]) continue;
        for (size_t i = 0; i < kPlane; ++i) plane[i] = src[t][4 * i]; // .r
        GRV_HIP(e, hipMemcpy(e->d_noise + t * kPlane, plane.data(), kPlane, hipMemcpyHostToDevice));
    }
    return GRV_OK;
}


int grv_render_frame_wgsl(grv_engine *e, const GrvWgslParams *p, float *d_rgba, uint32_t *d_steps,
                          uint64_t *total_steps, void *stream) {
    if (!e) return GRV_ERR_INVALID;
    if (!p || !d_rgba) return fail(e, GRV_ERR_INVALID, "null argument");
    if (p->arith != GRV_ARITH_STRICT && p->arith != GRV_ARITH_FAST)
        return fail(e, GRV_ERR_INVALID, "invalid arith %d", p->arith);
    WgslParams P{};
    std::memcpy(P.inv_view, p->inv_view, sizeof P.inv_view);
    std::memcpy(P.inv_proj, p->inv_proj, sizeof P.inv_proj);
    std::memcpy(P.position, p->position, sizeof P.position);
    P.mass = p->mass;
    P.spin = p->spin;
    P.jitter[0] = p->jitter[0];
    P.jitter[1] = p->jitter[1];
    P.max_steps = p->max_steps;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return run_shader_frame(e, p->width, p->height, p->tile_world, p->tile_rank, total_steps, s,
                            [&](const FrameGeom &G, uint32_t n, unsigned long long *tot) {
                                return p->arith == GRV_ARITH_FAST
                                           ? launch_wgsl_symplectic_fast(G, P, d_rgba, d_steps, tot, n, s)
                                           : launch_wgsl_symplectic(G, P, d_rgba, d_steps, tot, n, s);
                            });
}

int grv_render_frame_glsl(grv_engine *e, const GrvGlslParams *p, float *d_rgba, uint32_t *d_steps,
                          uint64_t *total_steps, void *stream) {
    if (!e) return GRV_ERR_INVALID;
    if (!p || !d_rgba) return fail(e, GRV_ERR_INVALID, "null argument");
    if (p->arith != GRV_ARITH_STRICT && p->arith != GRV_ARITH_FAST)
        return fail(e, GRV_ERR_INVALID, "invalid arith %d", p->arith);
    GlslParams P{};
    P.mass = p->mass;
    P.spin = p->spin;
    P.zoom = p->zoom;
    P.mouse[0] = p->mouse[0];
    P.mouse[1] = p->mouse[1];
    P.disk_size = p->disk_size;
    P.disk_scale_height = p->disk_scale_height;
    P.disk_density = p->disk_density;
    P.disk_temp = p->disk_temp;
    P.lensing_strength = p->lensing_strength;
    P.time = p->time;
    P.turbulence = p->turbulence;
    P.max_ray_steps = p->max_ray_steps;
    P.tone_map = p->tone_map;
    P.features = p->features;
    P.quality = p->quality;
    P.show_redshift = p->show_redshift;
    P.show_kerr_shadow = p->show_kerr_shadow;
    P.debug = p->debug;
    std::memcpy(P.cam_pos, p->cam_pos, sizeof P.cam_pos);
    std::memcpy(P.cam_quat, p->cam_quat, sizeof P.cam_quat);
    P.shadow_count = p->shadow_count;
    std::memcpy(P.shadow_curve, p->shadow_curve, sizeof P.shadow_curve);
    if (!e->d_noise) { // first GLSL frame of this engine: the seeded default textures
        std::vector<uint8_t> a(256 * 256 * 4), b(256 * 256 * 4);
        grv_seeded_noise_rgba8(1u, 256, a.data());
        grv_seeded_noise_rgba8(2u, 256, b.data());
        int rc = grv_set_glsl_noise(e, a.data(), b.data());
        if (rc != GRV_OK) return rc;
    }
    P.noise_r = e->d_noise;
    P.blue_r = e->d_noise + 256 * 256;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return run_shader_frame(e, p->width, p->height, p->tile_world, p->tile_rank, total_steps, s,
                            [&](const FrameGeom &G, uint32_t n, unsigned long long *tot) {
                                return p->arith == GRV_ARITH_FAST
                                           ? launch_glsl_fragment_fast(G, P, d_rgba, d_steps, tot, n, s)
                                           : launch_glsl_fragment(G, P, d_rgba, d_steps, tot, n, s);
                            });
}

float grv_taa_effective_blend(float blend_factor, float v) {
    if (v > 0.001f) return std::fmax(0.05f, std::fmin(0.9f, 0.9f - v * 6.0f));
    return blend_factor;
}

int grv_post_taa_resolve(grv_engine *e, const GrvTaaParams *p, const float *d_current,
                         const float *d_history, float *d_out, void *stream) {
    if (!e) return GRV_ERR_INVALID;
    if (!p || !d_current || !d_history || !d_out) return fail(e, GRV_ERR_INVALID, "null argument");
    if (d_out == d_current || d_out == d_history) return fail(e, GRV_ERR_INVALID, "taa: out aliases an input");
    GRV_HIP(e, hipSetDevice(e->device));
    GRV_HIP(e, launch_taa_resolve(p->width, p->height, d_current, d_history, p->blend_factor,
                                  p->camera_moving, p->half_storage, d_out, static_cast<hipStream_t>(stream)));
    return GRV_OK;
}

int grv_post_ataa_resolve(grv_engine *e, const GrvAtaaParams *p, const float *d_current,
                          const float *d_history, float *d_out, void *stream) {
    if (!e) return GRV_ERR_INVALID;
    if (!p || !d_current || !d_history || !d_out) return fail(e, GRV_ERR_INVALID, "null argument");
    if (d_out == d_current || d_out == d_history) return fail(e, GRV_ERR_INVALID, "ataa: out aliases an input");
    AtaaCameraHost cam;
    std::memcpy(cam.inv_view, p->inv_view, sizeof cam.inv_view);
    std::memcpy(cam.inv_proj, p->inv_proj, sizeof cam.inv_proj);
    std::memcpy(cam.prev_view_proj, p->prev_view_proj, sizeof cam.prev_view_proj);
    std::memcpy(cam.position, p->position, sizeof cam.position);
    GRV_HIP(e, hipSetDevice(e->device));
    GRV_HIP(e, launch_ataa_resolve(p->width, p->height, cam, d_current, d_history, p->half_storage, d_out,
                                   static_cast<hipStream_t>(stream)));
    return GRV_OK;
}

void grv_bloom_params_default(uint32_t width, uint32_t height, GrvBloomParams *p) {
    if (!p) return;
    p->width = width;
    p->height = height;
    p->intensity = 0.5f; // bloom.ts:34-39
    p->threshold = 0.8f;
    p->blur_passes = 2;
    p->half_storage = 1;
}

int grv_post_bloom(grv_engine *e, const GrvBloomParams *p, const float *d_scene, float *d_out,
                   void *stream) {
    if (!e) return GRV_ERR_INVALID;
    if (!p || !d_scene || !d_out) return fail(e, GRV_ERR_INVALID, "null argument");
    if (p->blur_passes < 0 || p->blur_passes > 64) return fail(e, GRV_ERR_INVALID, "blur_passes out of range");
    if (d_out == d_scene) return fail(e, GRV_ERR_INVALID, "bloom: out aliases the scene");
    GRV_HIP(e, hipSetDevice(e->device));
    const size_t need = bloom_scratch_floats(p->width, p->height) * sizeof(float);
    if (need > e->post_bytes) {
        if (e->post_mem) (void)hipFree(e->post_mem);
    if (e->rt.mem) (void)hipFree(e->rt.mem);
        e->post_mem = nullptr;
        e->post_bytes = 0;
        GRV_HIP(e, hipMalloc(&e->post_mem, need));
        e->post_bytes = need;
    }
    GRV_HIP(e, launch_bloom(p->width, p->height, d_scene, p->threshold, p->intensity, p->blur_passes,
                            p->half_storage, static_cast<float *>(e->post_mem), d_out,
                            static_cast<hipStream_t>(stream)));
    return GRV_OK;
}

// ---- renderer layer ----
namespace {
int ensure_targets(grv_engine *e, uint32_t w, uint32_t h, hipStream_t s) {
    if (e->rt.mem && e->rt.w == w && e->rt.h == h) return GRV_OK;
    if (e->rt.mem) (void)hipFree(e->rt.mem);
    e->rt = grv_engine::Targets{};
    const size_t bytes = (size_t)3 * w * h * 4 * sizeof(float);
    GRV_HIP(e, hipMalloc(reinterpret_cast<void **>(&e->rt.mem), bytes));
    GRV_HIP(e, hipMemsetAsync(e->rt.mem, 0, bytes, s)); // textures start zeroed
    e->rt.w = w;
    e->rt.h = h;
    return GRV_OK;
}
int ensure_bloom_scratch(grv_engine *e, uint32_t w, uint32_t h, hipStream_t s) {
    const size_t need = bloom_scratch_floats(w, h) * sizeof(float);
    if (need <= e->post_bytes) return GRV_OK;
    if (e->post_mem) (void)hipFree(e->post_mem);
    e->post_mem = nullptr;
    e->post_bytes = 0;
    GRV_HIP(e, hipMalloc(&e->post_mem, need));
    GRV_HIP(e, hipMemsetAsync(e->post_mem, 0, need, s));
    e->post_bytes = need;
    return GRV_OK;
}
// halton(index, base), compute.wgsl.ts:134-145, in f32
float halton_f32(uint32_t index, uint32_t base) {
    float result = 0.0f, f = 1.0f / (float)base;
    for (uint32_t i = index; i > 0u; i /= base) {
        result += f * (float)(i % base);
        f = f / (float)base;
    }
    return result;
}
} // namespace

void grv_renderer_reset(grv_engine *e) {
    if (!e) return;
    if (e->rt.mem) (void)hipFree(e->rt.mem);
    e->rt = grv_engine::Targets{};
}
uint32_t grv_renderer_frame_count(const grv_engine *e) { return e ? e->rt.frames : 0u; }

int grv_webgpu_render(grv_engine *e, const float *cu, const float *pp, int32_t max_steps, int32_t arith,
                      float *d_screen, void *stream) {
    if (!e) return GRV_ERR_INVALID;
    if (!cu || !pp || !d_screen) return fail(e, GRV_ERR_INVALID, "null argument");
    const uint32_t w = (uint32_t)pp[2], h = (uint32_t)pp[3]; // u32(physics.resolution), compute.wgsl.ts:149-150
    if (w == 0 || h == 0 || (uint64_t)w * h > (1ull << 27)) return fail(e, GRV_ERR_INVALID, "bad resolution");
    hipStream_t s = static_cast<hipStream_t>(stream);
    GRV_HIP(e, hipSetDevice(e->device));
    int rc = ensure_targets(e, w, h, s);
    if (rc != GRV_OK) return rc;
    const size_t plane = (size_t)w * h * 4;
    float *compute_tex = e->rt.mem, *hist[2] = {e->rt.mem + plane, e->rt.mem + 2 * plane};
    // Pass 1: main ray march.  CameraUniforms floats: inv_view 32..47, inv_proj 48..63,
    // prev_view_proj 64..79, position 80..82 (types/webgpu.ts:95-116)
    GrvWgslParams wp;
    std::memset(&wp, 0, sizeof wp);
    wp.width = w;
    wp.height = h;
    std::memcpy(wp.inv_view, cu + 32, sizeof wp.inv_view);
    std::memcpy(wp.inv_proj, cu + 48, sizeof wp.inv_proj);
    std::memcpy(wp.position, cu + 80, sizeof wp.position);
    wp.mass = pp[0];
    wp.spin = pp[1];
    const uint32_t fi = e->rt.frames; // paramsWithFrame.frameIndex = this.frameCount
    wp.jitter[0] = halton_f32((fi % 8u) + 1u, 2u) - 0.5f;
    wp.jitter[1] = halton_f32((fi % 8u) + 1u, 3u) - 0.5f;
    wp.max_steps = max_steps > 0 ? max_steps : 150;
    wp.tile_world = 1;
    wp.arith = arith;
    rc = grv_render_frame_wgsl(e, &wp, compute_tex, nullptr, nullptr, stream);
    if (rc != GRV_OK) return rc;
    GRV_HIP(e, launch_post_quantize(compute_tex, w * h, s)); // texture_storage_2d<rgba16float>
    // Pass 2: ATAA resolve, history ping-pong (renderer.ts:319-345, 385-395)
    const uint32_t hi = e->rt.hist, nx = 1u - hi;
    AtaaCameraHost cam;
    std::memcpy(cam.inv_view, cu + 32, sizeof cam.inv_view);
    std::memcpy(cam.inv_proj, cu + 48, sizeof cam.inv_proj);
    std::memcpy(cam.prev_view_proj, cu + 64, sizeof cam.prev_view_proj);
    std::memcpy(cam.position, cu + 80, sizeof cam.position);
    GRV_HIP(e, launch_ataa_resolve(w, h, cam, compute_tex, hist[hi], 1, hist[nx], s));
    // Pass 3: blit with Reinhard (renderer.ts:14-50, 397-411)
    GRV_HIP(e, launch_blit_reinhard(w, h, hist[nx], d_screen, s));
    e->rt.hist = nx;
    e->rt.frames++;
    return GRV_OK;
}

int grv_webgl_render(grv_engine *e, const GrvGlslParams *p, int32_t bloom_enabled, int32_t camera_moving,
                     float *d_screen, void *stream) {
    if (!e) return GRV_ERR_INVALID;
    if (!p || !d_screen) return fail(e, GRV_ERR_INVALID, "null argument");
    const uint32_t w = p->width, h = p->height;
    if (w == 0 || h == 0 || (uint64_t)w * h > (1ull << 27)) return fail(e, GRV_ERR_INVALID, "bad resolution");
    if (p->tile_world > 1) return fail(e, GRV_ERR_INVALID, "the renderer layer draws whole frames");
    hipStream_t s = static_cast<hipStream_t>(stream);
    GRV_HIP(e, hipSetDevice(e->device));
    int rc = ensure_targets(e, w, h, s);
    if (rc != GRV_OK) return rc;
    rc = ensure_bloom_scratch(e, w, h, s);
    if (rc != GRV_OK) return rc;
    const size_t plane = (size_t)w * h * 4;
    float *scene = e->rt.mem, *ping = e->rt.mem + plane, *pong = e->rt.mem + 2 * plane;
    // scene pass into the RGBA16F scene target, linear output (manager.ts:84-86)
    GrvGlslParams gp = *p;
    gp.tone_map = 0;
    rc = grv_render_frame_glsl(e, &gp, scene, nullptr, nullptr, stream);
    if (rc != GRV_OK) return rc;
    GRV_HIP(e, launch_post_quantize(scene, w * h, s));
    // ReprojectionManager.resolve: write index 0 -> write pong, read ping (reprojection.ts:209-216)
    float *write_tex = e->rt.hist == 0 ? pong : ping;
    const float *read_tex = e->rt.hist == 0 ? ping : pong;
    GRV_HIP(e, launch_taa_resolve(w, h, scene, read_tex, 0.75f, camera_moving, 1, write_tex, s));
    e->rt.hist = 1u - e->rt.hist;
    // bloom (features.bloom) or plain presentation: both are the combine pass (bloom.ts:443-632)
    float *scratch = static_cast<float *>(e->post_mem);
    if (bloom_enabled) {
        GRV_HIP(e, launch_bloom(w, h, write_tex, 0.8f, 0.5f, 2, 1, scratch, d_screen, s));
    } else {
        // drawTextureToScreen: combine with intensity 0; the bloom input is a stale dummy upstream,
        // here the (zero or last) bright-pass target, multiplied by 0 either way
        GRV_HIP(e, launch_bloom(w, h, write_tex, 3.0e38f, 0.0f, 0, 1, scratch, d_screen, s));
    }
    e->rt.frames++;
    return GRV_OK;
}

int grv_webgpu_render_host(grv_engine *e, const float *cu, const float *pp, int32_t max_steps,
                           int32_t arith, float *screen) {
    if (!e) return GRV_ERR_INVALID;
    if (!cu || !pp || !screen) return fail(e, GRV_ERR_INVALID, "null argument");
    const size_t bytes = (size_t)(uint32_t)pp[2] * (uint32_t)pp[3] * 4 * sizeof(float);
    if (bytes == 0 || bytes > ((size_t)1 << 31)) return fail(e, GRV_ERR_INVALID, "bad resolution");
    GRV_HIP(e, hipSetDevice(e->device));
    int rc = ensure_stage(e, bytes);
    if (rc != GRV_OK) return rc;
    rc = grv_webgpu_render(e, cu, pp, max_steps, arith, static_cast<float *>(e->stage_mem), nullptr);
    if (rc != GRV_OK) return rc;
    GRV_HIP(e, hipDeviceSynchronize());
    GRV_HIP(e, hipMemcpy(screen, e->stage_mem, bytes, hipMemcpyDeviceToHost));
    return GRV_OK;
}

int grv_webgl_render_host(grv_engine *e, const GrvGlslParams *p, int32_t bloom_enabled,
                          int32_t camera_moving, float *screen) {
    if (!e) return GRV_ERR_INVALID;
    if (!p || !screen) return fail(e, GRV_ERR_INVALID, "null argument");
    const size_t bytes = (size_t)p->width * p->height * 4 * sizeof(float);
    if (bytes == 0 || bytes > ((size_t)1 << 31)) return fail(e, GRV_ERR_INVALID, "bad resolution");
    GRV_HIP(e, hipSetDevice(e->device));
    int rc = ensure_stage(e, bytes);
    if (rc != GRV_OK) return rc;
    rc = grv_webgl_render(e, p, bloom_enabled, camera_moving, static_cast<float *>(e->stage_mem), nullptr);
    if (rc != GRV_OK) return rc;
    GRV_HIP(e, hipDeviceSynchronize());
    GRV_HIP(e, hipMemcpy(screen, e->stage_mem, bytes, hipMemcpyDeviceToHost));
    return GRV_OK;
}

int grv_unpack_tiles_device(grv_engine *e, const GrvRenderParams *p, uint32_t rank,
                            const void *d_packed, void *d_image, size_t bpp, void *stream) {
    if (!e) return GRV_ERR_INVALID;
    if (!p || !d_packed || !d_image || bpp == 0 || (bpp & 3u)) return fail(e, GRV_ERR_INVALID, "bad unpack request");
    GrvRenderParams q = *p;
    q.tile_rank = rank;
    FrameGeom G;
    frame_geometry(q, G);
    GRV_HIP(e, hipSetDevice(e->device));
    GRV_HIP(e, launch_unpack_tiles(G, d_packed, d_image, (uint32_t)(bpp / 4), static_cast<hipStream_t>(stream)));
    return GRV_OK;
}

void grv_camera_look_at(const double eye[3], const double target[3], const double up[3],
                        double fovy_rad, double aspect, GrvCamera *cam) {
    if (!cam) return;
    auto norm = [](double v[3]) {
        const double len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        v[0] /= len;
        v[1] /= len;
        v[2] /= len;
    };
    double z[3] = {eye[0] - target[0], eye[1] - target[1], eye[2] - target[2]};
    norm(z);
    double x[3] = {up[1] * z[2] - up[2] * z[1], up[2] * z[0] - up[0] * z[2], up[0] * z[1] - up[1] * z[0]};
    norm(x);
    const double y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
    std::memset(cam, 0, sizeof *cam);
    for (int k = 0; k < 3; ++k) {
        cam->position[k] = eye[k];
        cam->inv_view[0 + k] = x[k];
        cam->inv_view[4 + k] = y[k];
        cam->inv_view[8 + k] = z[k];
        cam->inv_view[12 + k] = eye[k];
    }
    cam->inv_view[15] = 1.0;
    const double near = 0.1, far = 1000.0; // WebGPUCanvas.tsx:151
    const double f = 1.0 / std::tan(fovy_rad / 2.0);
    const double a = f / aspect, b = f;
    const double c = (far + near) / (near - far);
    const double d = 2.0 * far * near / (near - far);
    cam->inv_proj[0] = 1.0 / a;
    cam->inv_proj[5] = 1.0 / b;
    cam->inv_proj[11] = 1.0 / d;
    cam->inv_proj[14] = -1.0;
    cam->inv_proj[15] = c / d;
    cam->pixel_offset[0] = 0.5;
    cam->pixel_offset[1] = 0.5;
}

void grv_camera_from_uniforms(const float *u, GrvCamera *cam) {
    if (!u || !cam) return;
    // src/types/webgpu.ts:95-116: view 0, proj 16, inv_view 32, inv_proj 48, prev 64, position 80
    for (int k = 0; k < 16; ++k) {
        cam->inv_view[k] = u[32 + k];
        cam->inv_proj[k] = u[48 + k];
    }
    for (int k = 0; k < 3; ++k) cam->position[k] = u[80 + k];
    cam->pixel_offset[0] = 0.0;
    cam->pixel_offset[1] = 0.0;
}

int grv_generate_spectrum_lut_device(grv_engine *e, size_t width, size_t height, double max_temp,
                                     float *d_out, void *stream) {
    if (!e) return GRV_ERR_INVALID;
    if (!d_out || width == 0 || height == 0 || width * height > (1ull << 26))
        return fail(e, GRV_ERR_INVALID, "bad LUT request");
    GRV_HIP(e, hipSetDevice(e->device));
    GRV_HIP(e, launch_spectrum_lut(d_out, (uint32_t)width, (uint32_t)height, max_temp,
                                   static_cast<hipStream_t>(stream)));
    return GRV_OK;
}

int grv_generate_spectrum_lut(grv_engine *e, size_t width, size_t height, double max_temp,
                              float *out_host) {
    if (!e) return GRV_ERR_INVALID;
    if (!out_host || width == 0 || height == 0 || width * height > (1ull << 26))
        return fail(e, GRV_ERR_INVALID, "bad LUT request");
    GRV_HIP(e, hipSetDevice(e->device));
    const size_t bytes = width * height * 4 * sizeof(float);
    int rc = ensure_stage(e, bytes);
    if (rc != GRV_OK) return rc;
    rc = grv_generate_spectrum_lut_device(e, width, height, max_temp, static_cast<float *>(e->stage_mem), nullptr);
    if (rc != GRV_OK) return rc;
    GRV_HIP(e, hipDeviceSynchronize());
    GRV_HIP(e, hipMemcpy(out_host, e->stage_mem, bytes, hipMemcpyDeviceToHost));
    return GRV_OK;
}

const float *grv_get_sab_ptr(const grv_engine *e) { return e ? e->sab.data() : nullptr; }

int grv_attach_sab(grv_engine *e, float *ptr) {
    if (!e) return GRV_ERR_INVALID;
    e->sab_ext = ptr;
    return GRV_OK;
}

void grv_set_camera_state(grv_engine *e, double px, double py, double pz) {
    if (!e) return;
    e->camera.position[0] = px;
    e->camera.position[1] = py;
    e->camera.position[2] = pz;
}

void grv_set_auto_spin(grv_engine *e, int enabled) {
    if (e) e->camera.auto_spin = enabled != 0;
}

int grv_tick_sab(grv_engine *e, double dt_override) {
    if (!e) return GRV_ERR_INVALID;
    tick_sab_host(e->sab_block(), e->mass, e->spin, e->spin_c, event_horizon(e->mass, e->spin_c),
                  isco_prograde(e->mass, e->spin_c), e->camera, e->last_good_camera, dt_override);
    return GRV_OK;
}

double grv_compute_disk_flux(const grv_engine *e, double r) {
    return page_thorne_flux_host(r, e->mass, e->spin_c, 1.0);
}

double grv_compute_shadow_radius(const grv_engine *e) { return schwarzschild_shadow_radius_host(e->mass); }

size_t grv_compute_shadow_curve(const grv_engine *e, double theta_obs, size_t n_points, float *out) {
    if (!e || !out) return 0;
    const std::vector<double> c = bardeen_shadow_host(e->mass, e->spin_c, theta_obs, n_points);
    for (size_t i = 0; i < c.size(); ++i) out[i] = (float)c[i];
    return c.size() / 2;
}

int grv_compute_shadow_shift(const grv_engine *e, double theta_obs, float out2[2]) {
    if (!e || !out2) return GRV_ERR_INVALID;
    const std::vector<double> c = bardeen_shadow_host(e->mass, e->spin_c, theta_obs, 32);
    double lo = 0.0, hi = 0.0;
    if (!c.empty()) {
        lo = hi = c[0];
        for (size_t i = 0; i < c.size(); i += 2) {
            lo = c[i] < lo ? c[i] : lo;
            hi = c[i] > hi ? c[i] : hi;
        }
    }
    out2[0] = (float)lo;
    out2[1] = (float)hi;
    return GRV_OK;
}

int grv_generate_disk_lut(grv_engine *e, float *out512) {
    if (!e) return GRV_ERR_INVALID;
    if (!out512) return fail(e, GRV_ERR_INVALID, "null output");
    GRV_HIP(e, hipSetDevice(e->device));
    const uint32_t w = 512; // lut_width, lib.rs:65
    int rc = ensure_stage(e, 4096 + w * sizeof(double));
    if (rc != GRV_OK) return rc;
    float *d_out = static_cast<float *>(e->stage_mem);
    double *d_tmp = reinterpret_cast<double *>(static_cast<char *>(e->stage_mem) + 4096);
    GRV_HIP(e, launch_disk_temperature_lut(d_out, d_tmp, w, e->mass, e->spin_c, nullptr));
    GRV_HIP(e, hipDeviceSynchronize());
    GRV_HIP(e, hipMemcpy(out512, d_out, w * sizeof(float), hipMemcpyDeviceToHost));
    std::memcpy(e->disk_lut.data(), out512, w * sizeof(float)); // self.lut_buffer = ... (lib.rs:108)
    return GRV_OK;
}

const float *grv_get_disk_lut_ptr(const grv_engine *e) { return e ? e->disk_lut.data() : nullptr; }

// ---- spacetime read-outs (spacetime_viz.hip) ----

double grv_compute_kretschner(const grv_engine *e, double r, double theta) {
    return e ? viz_kretschner(viz_hole(e), r, theta) : NAN;
}
double grv_compute_light_cone_tilt(const grv_engine *e, double r, double theta) {
    return e ? viz_light_cone_tilt(viz_hole(e), r, theta) : NAN;
}
double grv_compute_frame_drag_omega(const grv_engine *e, double r, double theta) {
    return e ? viz_frame_drag_omega(viz_hole(e), r, theta) : NAN;
}
double grv_compute_flamm_height(const grv_engine *e, double r) {
    return e ? viz_flamm_height(r, e->mass) : NAN;
}
double grv_compute_proper_distance(const grv_engine *e, double r1, double r2, size_t n_steps) {
    return e ? viz_proper_distance(viz_hole(e), r1, r2, n_steps) : NAN;
}

int grv_generate_field(grv_engine *e, int field, double r_min, double r_max, size_t n_radial,
                       size_t n_polar, float *out) {
    if (e && (field < GRV_FIELD_CURVATURE || field > GRV_FIELD_FRAME_DRAG))
        return fail(e, GRV_ERR_INVALID, "unknown field %d", field);
    return viz_grid(e, n_radial, n_polar, out, [&](float *d) {
        return launch_viz_field(field, viz_hole(e), r_min, r_max, (uint32_t)n_radial,
                                (uint32_t)n_polar, d, nullptr);
    });
}

int grv_generate_embedding_mesh(grv_engine *e, double r_min, double r_max, size_t n_radial,
                                size_t n_angular, float *out) {
    return viz_grid(e, n_radial, n_angular, out, [&](float *d) {
        return launch_embedding_mesh(viz_hole(e), r_min, r_max, (uint32_t)n_radial,
                                     (uint32_t)n_angular, d, nullptr);
    });
}

int grv_generate_ergosphere_mesh(grv_engine *e, size_t n_polar, size_t n_azimuthal, float *out) {
    return viz_grid(e, n_polar, n_azimuthal, out, [&](float *d) {
        return launch_ergosphere_mesh(viz_hole(e), (uint32_t)n_polar, (uint32_t)n_azimuthal, d,
                                      nullptr);
    });
}

void grv_get_sab_layout(size_t out5[5]) {
    if (!out5) return;
    out5[0] = kOffControl;
    out5[1] = kOffCamera;
    out5[2] = kOffPhysics;
    out5[3] = kOffTelemetry;
    out5[4] = kOffLuts;
}

} // extern "C"
