// engine.hip -- lifecycle, closed forms, batch / single-ray / frame entry points of the C ABI
// (include/gravitas_abi.h) on top of the segment kernels.  See engine_internal.hpp.
#include "engine_internal.hpp"

#include <atomic>
#include "strict_libm.hpp"

#include <chrono>

namespace grvhost {

// ---------------------------------------------------------------------------
// closed forms (host scalars; gravitas-core/src/metric/{mod,kerr}.rs)
// ---------------------------------------------------------------------------

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
    const double z1 = 1.0 + strictm::sl_pow(1.0 - a2, 1.0 / 3.0) *
                                (strictm::sl_pow(1.0 + a_star, 1.0 / 3.0) + strictm::sl_pow(1.0 - a_star, 1.0 / 3.0));
    const double z2 = std::sqrt(3.0 * a2 + z1 * z1);
    const double disc = (3.0 - z1) * (3.0 + z1 + 2.0 * z2);
    const double root = disc < 0.0 ? 0.0 : std::sqrt(disc);
    return m * (3.0 + z2 - root);
}
// metric/kerr.rs:91-94
double photon_sphere(double m, double a_star) {
    const double term = (2.0 / 3.0) * strictm::sl_acos(-a_star);
    return 2.0 * m * (1.0 + strictm::sl_cos(term));
}
// metric/kerr.rs:181-189 on covariant_bl g_tt (kerr.rs:241-254), theta = pi/2;
// gravitas-wasm/src/lib.rs:97-105
double dilation(double m, double a_star, double r) {
    const double a = a_star * m;
    const double theta = 1.57079632679489661923;
    const double cos_theta = strictm::sl_cos(theta);
    const double sigma = r * r + a * a * (cos_theta * cos_theta);
    const double g_tt = -(1.0 - (2.0 * m * r) / sigma);
    const double td = g_tt >= 0.0 ? 0.0 : std::sqrt(-g_tt);
    return td <= 0.0 ? 100.0 : 1.0 / td;
}
// physics/redshift.rs:65-95
double g_factor(double r, double mass, double spin, double lambda) {
    const double a = spin * mass, r2 = r * r, a2 = a * a, m = mass;
    const double omega = std::sqrt(m) / (strictm::sl_pow(r, 1.5) + a * std::sqrt(m)); // as on the device
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


// ---------------------------------------------------------------------------
// shared host helpers
// ---------------------------------------------------------------------------
int fail(grv_engine *e, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (e) e->err = buf;
    return code;
}



size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Takes the workspace set for a frame / batch of `slots` rays queued on `s` (allocating or growing
// it) and orders `s` behind the set's previous user.  The previous call's set is taken again unless
// that call sits on another stream: only then does the other set come into play (two frames in
// flight); a one-stream caller never allocates it.
int ensure_workspace(grv_engine *e, size_t slots, hipStream_t s) {
    int pick = e->wlast;
    if (e->ctl_stream && s == e->ctl_stream) {
        pick = 2; // the control stream's own set: a worker-sized batch never waits for a frame's workspace
    } else {
        if (e->wset[pick].used && e->wset[pick].last_stream != s) pick ^= 1;
        e->wlast = pick;
    }
    grv_engine::WorkSet &W = e->wset[pick];
    GRV_HIP(e, hipSetDevice(e->device));
    if (!W.done) GRV_HIP(e, hipEventCreateWithFlags(&W.done, hipEventDisableTiming));
    e->cur = &W;
    if (!(slots <= W.slots && W.mem)) {
        if (W.mem) {
            (void)hipFree(W.mem); // implicit device synchronise: no frame still uses it
            W.mem = nullptr;
            W.slots = 0;
            W.bytes = 0;
            W.used = false;
        }
        // 10 f64 components + crossing records + 3 u32 + two live lists, each 256-B aligned, + counters
        const size_t cap = align_up(slots, 64);
        const size_t f64b = align_up(cap * sizeof(double), 256);
        const size_t u32b = align_up(cap * sizeof(uint32_t), 256);
        const size_t total = f64b * (10 + kMaxCrossRec) + u32b * 5 + 256;
        void *mem = nullptr;
        GRV_HIP(e, hipMalloc(&mem, total));
        W.bytes = total;
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
        W.live[0] = take32();
        W.live[1] = take32();
        W.d_counters = reinterpret_cast<uint32_t *>(p);
        // rc rows are addressed as rc[c * n + slot]
        W.mem = mem;
        W.slots = cap;
        W.ws = w;
    }
    // always ordered behind the set's previous user, same stream or not: a stream handle can be
    // re-issued by the runtime after its owner destroyed it, and a wait on the own stream is free
    if (W.used) GRV_HIP(e, hipStreamWaitEvent(s, W.done, 0));
    W.last_stream = s;
    W.ws.n = (uint32_t)slots;
    e->ws = W.ws;
    e->live[0] = W.live[0];
    e->live[1] = W.live[1];
    e->d_counters = W.d_counters;
    return GRV_OK;
}

// End of the call that took the current set: its later users wait for everything queued so far.
int release_workspace(grv_engine *e, hipStream_t s) {
    if (!e->cur) return GRV_OK;
    grv_engine::WorkSet *W = e->cur;
    e->cur = nullptr;
    W->used = true; // even if the record below fails: the next user then waits on the older event
    GRV_HIP(e, hipEventRecord(W->done, s));
    return GRV_OK;
}

int create_priority_stream(grv_engine *e, hipStream_t *out) {
    int least = 0, greatest = 0;
    GRV_HIP(e, hipDeviceGetStreamPriorityRange(&least, &greatest)); // numerically lower = higher priority
    GRV_HIP(e, hipStreamCreateWithPriority(out, hipStreamNonBlocking, greatest));
    return GRV_OK;
}

int control_stream(grv_engine *e, hipStream_t *out) {
    if (!e->ctl_stream) {
        const int rc = create_priority_stream(e, &e->ctl_stream);
        if (rc != GRV_OK) return rc;
    }
    *out = e->ctl_stream;
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
    P.final_launch = 0;
    P.try_cap = 0xFFFFFFFFu;
    P.shading = 0;
    P.disk_inner = 0.0;
    P.disk_outer = 0.0;
    P.max_crossings = 0xFFFFFFFFu;
    P.block_order = 0;
    return P;
}

bool options_valid(const GrvOptions &o) {
    if (o.method < GRV_METHOD_RKF45 || o.method > GRV_METHOD_SYMPLECTIC) return false;
    if (o.metric_kind < GRV_METRIC_KERR_BL || o.metric_kind > GRV_METRIC_SCHWARZSCHILD) return false;
    if (o.arith != GRV_ARITH_STRICT && o.arith != GRV_ARITH_FAST) return false;
    if (o.reserved != 0) return false;
    // STRICT takes any tolerance, as the reference does: 0 or NaN rejects every try and walks the
    // forced 1e-5 steps, a negative one accepts every try (integrator.rs:76-104).  The FAST
    // contract multiplies by 1 / tolerance and is defined for positive tolerances only.
    if (o.method == GRV_METHOD_RKF45 && o.arith == GRV_ARITH_FAST && !(o.tolerance > 0.0)) return false;
    // a NaN first step never completes a try (the reference's stepper loops forever on it,
    // integrator.rs:75-105): refused here rather than handed to a kernel that could not end
    if (o.method == GRV_METHOD_RKF45 && o.initial_step != o.initial_step) return false;
    return true;
}

hipError_t launch_segment(int arith, int kind, int method, const RayWorkspace &ws,
                          const SegmentParams &P, const uint32_t *live_in, uint32_t n_live,
                          uint32_t *live_out, uint32_t *cnt, hipStream_t s) {
    return arith == GRV_ARITH_FAST
               ? launch_segment_fast(kind, method, ws, P, live_in, n_live, live_out, cnt, s)
               : launch_segment_strict(kind, method, ws, P, live_in, n_live, live_out, cnt, s);
}

hipError_t launch_path(int arith, int kind, int method, const RayWorkspace &ws, const SegmentParams &P,
                       const double *states_in, double *paths, uint32_t *counts, uint32_t max_points,
                       hipStream_t s) {
    return arith == GRV_ARITH_FAST
               ? launch_path_fast(kind, method, ws, P, states_in, paths, counts, max_points, s)
               : launch_path_strict(kind, method, ws, P, states_in, paths, counts, max_points, s);
}

hipError_t launch_refill(int arith, int kind, int method, const RayWorkspace &ws,
                         const SegmentParams &P, uint32_t *cursor, int n_cu, hipStream_t s, uint32_t block_threads = 0) {
    return arith == GRV_ARITH_FAST ? launch_refill_fast(kind, method, ws, P, cursor, n_cu, s, block_threads)
                                   : launch_refill_strict(kind, method, ws, P, cursor, n_cu, s, block_threads);
}

// tries between two refill checks of a wave in the refill kernel
constexpr uint32_t kRefillPeriod = 8;

// Hard bound on the integrator tries one ray can take.  Every completed step costs a bounded number
// of tries: a reject shrinks |h| by >= 10 % (integrator.rs:97: factor max(0.9 ratio^-1/4, 0.1), ratio
// > 1), |h| <= 10 after the clamp, and below 1e-5 the forced step of integrator.rs:99-104 is taken
// unconditionally -- at most ln(1e6) / ln(1 / 0.9) = 132 rejects, one forced try.  steps <= max_steps,
// so 160 max_steps + 64 is never reached by a correct kernel; a ray still live there (a NaN the
// argument missed, a future stepper) is ended as TERM_MAXSTEPS instead of hanging the GPU.
uint32_t try_bound(const grv_engine *e, uint64_t max_steps) {
    // verification hook (grv_test_set_try_bound, an explicit call on this handle -- nothing in the
    // process environment reaches the result): a tiny bound makes the "never a hang" exit reachable
    if (e && e->try_bound_override) return e->try_bound_override;
    const uint64_t b = (max_steps > (0xFFFFFFFFull - 64u) / 160u) ? 0xFFFFFFFFull : 160u * max_steps + 64u;
    return (uint32_t)b;
}

// One integrate pass over the initialised workspace.
//   seg_tries == 0 (default): ONE launch that runs every ray to its end (or to try_bound): no live list.
//   seg_tries  > 0: the compacting wavefront schedule.  Every launch appends its survivors to the next live
//     list; the NEXT launch reads the list's length from device memory (integrate_compact_kernel), so the whole
//     pass -- L launches of seg_tries tries and a last one that runs whatever is left to its end -- is queued
//     at once.  L and the grids come from a forecast: the live counts the launches of the last completed pass
//     reported into pinned host memory (fire-and-forget stores; never waited for).  A wrong forecast costs time,
//     not correctness: grids stride over any count, and the last launch is unbounded.
//   Either way the host does not wait: the call returns with the work queued on `s`, and the results are
//   bitwise those of the single launch.
constexpr uint32_t kCompactMaxLaunches = 192;    // bounded launches of one pass (the last, unbounded one comes on top)
constexpr uint32_t kCompactUnknown = 0xFFFFFFFFu;
constexpr uint32_t kCompactTailRays = 196608u;   // fewer live rays than the chip holds at three waves per SIMD (3 072 waves): compaction has nothing left to fill, the last launch takes them

// share of the waves (the longest by the previous frame's count) that the compacting schedule runs in one unbounded
// launch beside the chain instead of through it: the chain's latency-bound tail -- a few thousand rays that outlive the
// bulk by hundreds of tries -- was 8 % of the frame (profiles/EXPERIMENTS.md section R)
// (A/B on one box, profiles/r06_ab_compact_head.jsonl: K = 16 at -12.0 % of the one-launch frame without a head start,
// -9 % with 1/32 of the waves, -7.0 % with 1/8, -5.8 % with 1/4; K = 64: -8.3 / -4 / -2.4 / -2.1 %.  An eighth: seven
// eighths of the frame still go through the chain.)
#ifndef GRV_COMPACT_HEAD_SHARE
#define GRV_COMPACT_HEAD_SHARE 8
#endif
constexpr uint32_t kCompactHeadShare = GRV_COMPACT_HEAD_SHARE;

int run_segments(grv_engine *e, const GrvOptions &o, SegmentParams P, uint32_t seg_tries,
                 hipStream_t s, bool profile, const uint32_t *head_order) {
    (void)profile; // the frame's ring events bracket the pass; nothing in here waits, so there are no host gaps in it
    const uint32_t bound = try_bound(e, o.max_steps);
    if (seg_tries == 0) {
        P.max_tries = bound;
        P.final_launch = 1;
        GRV_HIP(e, launch_segment(o.arith, o.metric_kind, o.method, e->ws, P, nullptr, e->ws.n,
                                  nullptr, nullptr, s));
        e->last_launches += 1;
        return GRV_OK;
    }
    const uint32_t n = e->ws.n;
    if (!e->compact_fb) {
        GRV_HIP(e, hipHostMalloc(reinterpret_cast<void **>(&e->compact_fb), (kCompactMaxLaunches + 2) * sizeof(uint32_t),
                                 hipHostMallocCoherent | hipHostMallocMapped));
        for (uint32_t j = 0; j < kCompactMaxLaunches + 2; ++j) e->compact_fb[j] = kCompactUnknown;
    }
    // forecast from what the launches of an earlier pass reported (entry j: live rays launch j started with); a pass
    // of another shape (ray count, tries per launch) says nothing about this one
    if (e->compact_fb_n != n || e->compact_fb_tries != seg_tries) {
        for (uint32_t j = 0; j < kCompactMaxLaunches + 2; ++j) e->compact_fb[j] = kCompactUnknown;
        e->compact_fb_n = n;
        e->compact_fb_tries = seg_tries;
    }
    volatile uint32_t *fb = e->compact_fb;
    uint32_t L = 0;
    for (uint32_t j = 1; j <= kCompactMaxLaunches; ++j) {
        const uint32_t v = fb[j];
        if (v == kCompactUnknown) { // nothing known from here on: a first pass, or rays outlived the last forecast
            const uint32_t guess = j == 1 ? (512u + seg_tries - 1u) / seg_tries : j + 7u;
            L = guess < kCompactMaxLaunches ? guess : kCompactMaxLaunches;
            break;
        }
        if (v < kCompactTailRays) {
            L = j;
            break;
        }
    }
    if (L == 0) L = kCompactMaxLaunches;
    auto blocks_for = [&](uint32_t j) {
        const uint32_t v = fb[j];
        uint64_t rays = n;
        if (v != kCompactUnknown) {
            // generous: surplus blocks cost microseconds, a grid that is too small serialises the launch
            rays = 2ull * v + 16384u;
            if (rays < 262144u) rays = 262144u;
            if (rays > n) rays = n;
        }
        // (one chunk per block when the forecast holds: a resident grid striding over the list measured 10 % slower,
        // as it did for the f32 marches -- the dispatcher refills a slot faster than a block turns around)
        return (uint32_t)((rays + 255u) / 256u);
    };
    uint32_t *c = e->d_counters + 4; // three rotating live counters (d_counters[0..3]: refill cursor and spares)
    GRV_HIP(e, hipMemsetAsync(c, 0, 3 * sizeof(uint32_t), s));
    // Head start: with a forecast (the previous frame's waves sorted longest-first) the top 1 / kCompactHeadShare of the
    // waves do not go through the chain at all -- they run to their end in ONE launch on the side stream, beside the
    // chain from its first launch on, the way the one-launch schedule runs them beside everything else.  The chain's
    // first list is then the rest (live[0][0 .. n_rest)); the head's list sits in the part of live[1] the chain never
    // appends to (its counts stay <= n_rest).  A stale forecast costs time, not correctness.
    uint32_t n_rest = n;
    bool head = false;
    if (head_order && kCompactHeadShare && (n & 63u) == 0u && n / 64u >= 4096u) {
        const uint32_t n_waves = n / 64u, n_head = n_waves / kCompactHeadShare;
        n_rest = (n_waves - n_head) * 64u;
        if (!e->sort_stream) GRV_HIP(e, hipStreamCreateWithFlags(&e->sort_stream, hipStreamNonBlocking));
        if (!e->head_from) GRV_HIP(e, hipEventCreateWithFlags(&e->head_from, hipEventDisableTiming));
        if (!e->head_done) GRV_HIP(e, hipEventCreateWithFlags(&e->head_done, hipEventDisableTiming));
        GRV_HIP(e, launch_split_order(head_order, n_waves, n_head, e->live[1] + n_rest, e->live[0], s));
        GRV_HIP(e, hipEventRecord(e->head_from, s));
        GRV_HIP(e, hipStreamWaitEvent(e->sort_stream, e->head_from, 0));
        SegmentParams H = P;
        H.max_tries = bound;
        H.final_launch = 1;
        H.order = nullptr;
        GRV_HIP(e, launch_segment(o.arith, o.metric_kind, o.method, e->ws, H, e->live[1] + n_rest, n_head * 64u, nullptr, nullptr,
                                  e->sort_stream));
        GRV_HIP(e, hipEventRecord(e->head_done, e->sort_stream));
        e->last_launches += 1;
        head = true;
    }
    GRV_HIP(e, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c), (int)n_rest, 1, s)); // launch 0 "reads" c[0]
    P.max_tries = seg_tries;
    P.final_launch = 0;
    P.order = nullptr;
    for (uint32_t j = 0; j <= L; ++j) { // launch 0: the identity list (null) or the split's rest, appends to live[1] / c[1]
        const bool last = j == L;
        if (last) {
            P.max_tries = bound;
            P.final_launch = 1;
        }
        uint32_t *out = last ? nullptr : e->live[(j + 1u) & 1u];
        const auto fn = o.arith == GRV_ARITH_FAST ? launch_compact_fast : launch_compact_strict;
        GRV_HIP(e, fn(o.metric_kind, o.method, e->ws, P, j == 0 ? (head ? e->live[0] : nullptr) : e->live[j & 1u], c + (j % 3u), out,
                      c + ((j + 1u) % 3u), c + ((j + 2u) % 3u), e->compact_fb + j, blocks_for(j), s));
    }
    if (head) GRV_HIP(e, hipStreamWaitEvent(s, e->head_done, 0)); // the finalize kernel reads every slot
    // what older passes reported beyond this pass's last launch is forgotten (if rays outlive the forecast, the next
    // pass finds "unknown" there and adds launches)
    for (uint32_t j = L + 1u; j <= kCompactMaxLaunches; ++j) fb[j] = kCompactUnknown;
    e->last_launches += L + 1u;
    return GRV_OK;
}

// Start of a frame / batch: take a counter block.  Accumulating callers keep adding to the current
// one; otherwise the call takes the other block and clears it on its own stream, ordered behind
// that block's previous user (whose finalize kernel may still be adding to it on another stream).
int begin_frame_stats(grv_engine *e, hipStream_t s) {
    e->stats_open = true;
    if (e->stats_accum) {
        // adds to the current block: only a clear of it (grv_frame_stats_reset on another stream)
        // has to come first
        if (e->stats_cleared_rec) GRV_HIP(e, hipStreamWaitEvent(s, e->stats_cleared, 0));
        return GRV_OK;
    }
    int b;
    if (e->ctl_stream && s == e->ctl_stream) {
        b = 2; // the control stream's own block (its previous user sits on the same stream)
    } else {
        e->stats_turn ^= 1;
        b = e->stats_turn;
    }
    e->d_stats = e->stats_blocks + b;
    if (e->stats_used[b]) GRV_HIP(e, hipStreamWaitEvent(s, e->stats_done[b], 0));
    if (e->stats_cleared_rec) GRV_HIP(e, hipStreamWaitEvent(s, e->stats_cleared, 0));
    GRV_HIP(e, hipMemsetAsync(e->d_stats, 0, sizeof(FrameStatsDev), s));
    for (float &m : e->last_ms) m = 0.f;
    e->last_launches = 0;
    e->ev_frames = 0; // unresolved events of earlier frames describe counters that are gone
    return GRV_OK;
}

int end_frame_stats(grv_engine *e, hipStream_t s) {
    if (!e->stats_open) return GRV_OK;
    e->stats_open = false;
    if (e->stats_accum) return GRV_OK; // accumulating calls on two streams must not order each other
    const int b = (int)(e->d_stats - e->stats_blocks);
    e->stats_used[b] = true;
    GRV_HIP(e, hipEventRecord(e->stats_done[b], s));
    return GRV_OK;
}

// Four events of the next profiled frame from the ring (created on demand).  When the ring is
// full the pending frames are resolved first (one synchronise every kEvRingFrames frames).
constexpr size_t kEvRingFrames = 1024;
int ring_events(grv_engine *e, hipEvent_t **ev4) {
    if (e->ev_frames >= kEvRingFrames) {
        const int rc = resolve_frame_events(e);
        if (rc != GRV_OK) return rc;
    }
    const size_t need = (e->ev_frames + 1) * 4;
    while (e->ev_ring.size() < need) {
        hipEvent_t ev;
        GRV_HIP(e, hipEventCreate(&ev));
        e->ev_ring.push_back(ev);
    }
    *ev4 = e->ev_ring.data() + e->ev_frames * 4;
    return GRV_OK;
}

// Adds the elapsed times of the pending profiled frames to last_ms (blocks until the last of
// their events has completed).  Called by grv_frame_stats, never by a frame call.
int resolve_frame_events(grv_engine *e) {
    if (e->ev_frames == 0) return GRV_OK;
    for (size_t f = 0; f < e->ev_frames; ++f) {
        hipEvent_t *q = e->ev_ring.data() + f * 4;
        float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
        GRV_HIP(e, hipEventSynchronize(q[3])); // frames may sit on different streams
        GRV_HIP(e, hipEventElapsedTime(&a, q[0], q[1]));
        GRV_HIP(e, hipEventElapsedTime(&b, q[1], q[2]));
        GRV_HIP(e, hipEventElapsedTime(&c, q[2], q[3]));
        GRV_HIP(e, hipEventElapsedTime(&d, q[0], q[3]));
        e->last_ms[0] += a;
        if (!(f < e->ev_loop.size() && e->ev_loop[f])) e->last_ms[1] += b;
        e->last_ms[3] += c;
        e->last_ms[4] += d;
    }
    e->ev_frames = 0;
    return GRV_OK;
}

int begin_march_order(grv_engine *e, int kind, uint32_t n_blocks, const uint32_t geom[4], hipStream_t s, MarchSched *out,
                      int *parity) {
    const int b = (int)(e->march_frames[kind]++ & 1u);
    grv_engine::MarchOrder &M = e->march_order[kind][b];
    if (!M.ready) GRV_HIP(e, hipEventCreateWithFlags(&M.ready, hipEventDisableTiming));
    // behind the sort (perhaps queued on another stream) that last wrote this parity's order
    if (M.ready_rec) GRV_HIP(e, hipStreamWaitEvent(s, M.ready, 0));
    if (M.n_blocks < n_blocks) {
        if (M.mem) {
            if (M.ready_rec) GRV_HIP(e, hipEventSynchronize(M.ready)); // its last user has finished before it is freed
            (void)hipFree(M.mem);
        }
        M.mem = nullptr;
        M.n_blocks = 0;
        M.has_order = false;
        M.ranked = false;
        GRV_HIP(e, hipMalloc(reinterpret_cast<void **>(&M.mem), (size_t)2 * n_blocks * sizeof(uint32_t)));
        M.n_blocks = n_blocks;
    }
    const uint32_t want[5] = {geom[0], geom[1], geom[2], geom[3], n_blocks};
    if (!M.has_order || std::memcmp(M.geom, want, sizeof want) != 0) {
        // first frame of exactly this geometry: natural order, no forecast yet
        M.has_order = false;
        M.ranked = false;
        GRV_HIP(e, launch_march_order_identity(M.mem + M.n_blocks, M.mem, n_blocks, s));
        // the table is in use on `s` from here on, whatever happens next: a later grow / free waits for it
        GRV_HIP(e, hipEventRecord(M.ready, s));
        M.ready_rec = true;
        std::memcpy(M.geom, want, sizeof want);
        M.has_order = true;
    }
    M.cur = n_blocks;
    out->cost = M.mem;
    out->order = M.mem + M.n_blocks;
    *parity = b;
    return GRV_OK;
}

int finish_march_order(grv_engine *e, int kind, int parity, hipStream_t s) {
    grv_engine::MarchOrder &M = e->march_order[kind][parity];
    // the next frame of this parity starts its longest blocks first
    hipStream_t q = s;
    if (kind == 2) { // beside the frame loop (the next frame reads the OTHER parity's order)
        if (!e->sort_stream) GRV_HIP(e, hipStreamCreateWithFlags(&e->sort_stream, hipStreamNonBlocking));
        if (!e->sort_from) GRV_HIP(e, hipEventCreateWithFlags(&e->sort_from, hipEventDisableTiming));
        GRV_HIP(e, hipEventRecord(e->sort_from, s));
        GRV_HIP(e, hipStreamWaitEvent(e->sort_stream, e->sort_from, 0));
        q = e->sort_stream;
    }
    GRV_HIP(e, launch_march_rank(M.mem, M.mem + M.n_blocks, M.cur, q));
    GRV_HIP(e, hipEventRecord(M.ready, q));
    M.ready_rec = true;
    M.ranked = true;
    return GRV_OK;
}

int ensure_lut(grv_engine *e, uint32_t w, uint32_t h, double tmax, hipStream_t s) {
    if (e->d_lut && e->lut_w == w && e->lut_h == h && e->lut_tmax == tmax) {
        GRV_HIP(e, hipStreamWaitEvent(s, e->lut_ready, 0)); // generated on another stream, perhaps
        return GRV_OK;
    }
    if (w == 0 || h == 0 || (uint64_t)w * h > (1ull << 26)) return fail(e, GRV_ERR_INVALID, "bad LUT shape");
    if (e->d_lut) (void)hipFree(e->d_lut); // implicit device synchronise: no queued frame still reads it
    e->d_lut = nullptr;
    GRV_HIP(e, hipMalloc(reinterpret_cast<void **>(&e->d_lut), (size_t)w * h * 4 * sizeof(float)));
    e->lut_w = e->lut_h = 0;
    GRV_HIP(e, launch_spectrum_lut(e->d_lut, w, h, tmax, s));
    GRV_HIP(e, hipEventRecord(e->lut_ready, s));
    e->lut_w = w;
    e->lut_h = h;
    e->lut_tmax = tmax;
    return GRV_OK;
}

// Tile grid of the image plane (row-major, physics-engine/_legacy_src/tiling.rs:38-56).  Tile ids run
// over a grid whose row pitch is the first integer >= ceil(width / 64) that is coprime with the
// number of ranks: dealing id k to rank k mod N then walks the ranks along a different diagonal in
// every row.  With the plain pitch a tile count per row that is a multiple of N (120 at 8K, 60 at 4K
// for N = 4) hands every rank whole tile COLUMNS, and the two ranks owning the columns through the
// hole's image carry its divergent waves alone (measured: 3.9 ms against 2.8 ms at 8K / 8 ranks).
// Ids in the pad column(s) hold no pixels.
// Device copy of generate_disk_lut's table for the current (mass, spin): physics/disk.rs:175-201
int ensure_disk_lut(grv_engine *e, hipStream_t s) {
    if (e->disk_lut_valid && e->disk_lut_mass == e->mass && e->disk_lut_spin == e->spin_c) {
        GRV_HIP(e, hipStreamWaitEvent(s, e->disk_lut_ready, 0));
        return GRV_OK;
    }
    constexpr size_t kScratchOff = 4096;
    if (!e->d_disk_lut)
        GRV_HIP(e, hipMalloc(reinterpret_cast<void **>(&e->d_disk_lut), kScratchOff + kDiskLutWidth * sizeof(double)));
    // regenerated in place (mass / spin changed): frames still queued on other streams read the old
    // table in their finalize kernels -- the rewrite goes behind the end of every workspace set's
    // last call
    for (auto &W : e->wset)
        if (W.used && &W != e->cur) GRV_HIP(e, hipStreamWaitEvent(s, W.done, 0));
    e->disk_lut_valid = false;
    double *scratch = reinterpret_cast<double *>(reinterpret_cast<char *>(e->d_disk_lut) + kScratchOff);
    GRV_HIP(e, launch_disk_temperature_lut(e->d_disk_lut, scratch, kDiskLutWidth, e->mass, e->spin_c, s));
    GRV_HIP(e, hipEventRecord(e->disk_lut_ready, s));
    e->disk_lut_mass = e->mass;
    e->disk_lut_spin = e->spin_c;
    e->disk_lut_valid = true;
    return GRV_OK;
}

uint32_t tile_pitch(uint32_t width, uint32_t world) {
    uint32_t p = (width + 63u) / 64u;
    if (world <= 1) return p;
    for (;; ++p) {
        uint32_t a = p, b = world;
        while (b) {
            const uint32_t t = a % b;
            a = b;
            b = t;
        }
        if (a == 1u) return p;
    }
}

void frame_geometry(const GrvRenderParams &p, FrameGeom &G) {
    G.width = p.width;
    G.height = p.height;
    G.tile_world = p.tile_world == 0 ? 1u : p.tile_world;
    G.tile_rank = p.tile_world == 0 ? 0u : p.tile_rank;
    G.tiles_x = tile_pitch(p.width, G.tile_world); // row pitch of the tile ids (>= tiles across)
    G.tiles_y = (p.height + 63u) / 64u;
    const uint32_t total = G.tiles_x * G.tiles_y;
    G.n_tiles_local = (total > G.tile_rank) ? (total - G.tile_rank + G.tile_world - 1u) / G.tile_world : 0u;
}

void stats_to_abi(const grv_engine *e, const FrameStatsDev &d, GrvFrameStats *out) {
    std::memset(out, 0, sizeof *out);
    out->rays = d.rays;
    out->accepted_steps = stats_total_steps(d);
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

} // namespace grvhost

using namespace grvhost;

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
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
    if (hipMalloc(reinterpret_cast<void **>(&e->stats_blocks), 3 * sizeof(FrameStatsDev)) != hipSuccess) return bail(GRV_ERR_OOM);
    if (hipMemset(e->stats_blocks, 0, 3 * sizeof(FrameStatsDev)) != hipSuccess) return bail(GRV_ERR_HIP);
    e->d_stats = e->stats_blocks;
    for (auto &ev : e->stats_done)
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return bail(GRV_ERR_HIP);
    if (hipEventCreateWithFlags(&e->lut_ready, hipEventDisableTiming) != hipSuccess) return bail(GRV_ERR_HIP);
    if (hipEventCreateWithFlags(&e->stats_cleared, hipEventDisableTiming) != hipSuccess) return bail(GRV_ERR_HIP);
    if (hipEventCreateWithFlags(&e->disk_lut_ready, hipEventDisableTiming) != hipSuccess) return bail(GRV_ERR_HIP);
    if (hipHostMalloc(reinterpret_cast<void **>(&e->h_counters), 4 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) return bail(GRV_ERR_OOM);
    if (hipHostMalloc(reinterpret_cast<void **>(&e->h_stats), sizeof(FrameStatsDev), hipHostMallocDefault) != hipSuccess) return bail(GRV_ERR_OOM);
    std::memset(e->h_stats, 0, sizeof(FrameStatsDev));
    // polled by the host while the kernel writes it: fine-grained (coherent) mapping spelled out, so
    // the poll does not depend on HIP_HOST_COHERENT
    if (hipHostMalloc(reinterpret_cast<void **>(&e->ray_out), sizeof(SingleRayOut),
                      hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) return bail(GRV_ERR_OOM);
    std::memset(e->ray_out, 0, sizeof(SingleRayOut));
    e->ev_ok = true;
    for (auto &ev : e->ev)
        if (hipEventCreate(&ev) != hipSuccess) e->ev_ok = false;
    *out = e;
    return GRV_OK;
}

void grv_engine_destroy(grv_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    // frames may still be queued on streams this handle has never seen (a device image's, a caller's): everything on the
    // device ends before the workspaces under it are released
    (void)hipDeviceSynchronize();
    for (auto &W : e->wset) {
        if (W.mem) (void)hipFree(W.mem);
        if (W.done) (void)hipEventDestroy(W.done);
    }
    if (e->stage_mem) (void)hipFree(e->stage_mem);
    for (auto &kind : e->march_order)
        for (auto &M : kind) {
            if (M.mem) (void)hipFree(M.mem);
            if (M.ready) (void)hipEventDestroy(M.ready);
        }
    if (e->path_stage) (void)hipHostFree(e->path_stage);
    if (e->d_lut) (void)hipFree(e->d_lut);
    if (e->d_disk_lut) (void)hipFree(e->d_disk_lut);
    if (e->d_noise) (void)hipFree(e->d_noise);
    if (e->post_mem) (void)hipFree(e->post_mem);
    if (e->rt.mem) (void)hipFree(e->rt.mem);
    if (e->stats_blocks) (void)hipFree(e->stats_blocks);
    for (auto &ev : e->stats_done)
        if (ev) (void)hipEventDestroy(ev);
    if (e->lut_ready) (void)hipEventDestroy(e->lut_ready);
    if (e->stats_cleared) (void)hipEventDestroy(e->stats_cleared);
    if (e->disk_lut_ready) (void)hipEventDestroy(e->disk_lut_ready);
    if (e->chain_done) (void)hipEventDestroy(e->chain_done);
    if (e->sort_stream) {
        (void)hipStreamSynchronize(e->sort_stream);
        (void)hipStreamDestroy(e->sort_stream);
    }
    if (e->sort_from) (void)hipEventDestroy(e->sort_from);
    if (e->head_from) (void)hipEventDestroy(e->head_from);
    if (e->head_done) (void)hipEventDestroy(e->head_done);
    if (e->h_counters) (void)hipHostFree(e->h_counters);
    if (e->compact_fb) (void)hipHostFree(e->compact_fb);
    if (e->h_stats) (void)hipHostFree(e->h_stats);
    if (e->ray_stream) {
        (void)hipStreamSynchronize(e->ray_stream);
        (void)hipStreamDestroy(e->ray_stream);
    }
    if (e->ctl_stream) {
        (void)hipStreamSynchronize(e->ctl_stream);
        (void)hipStreamDestroy(e->ctl_stream);
    }
    if (e->ray_out) (void)hipHostFree(e->ray_out);
    if (e->ev_ok)
        for (auto &ev : e->ev) (void)hipEventDestroy(ev);
    for (auto &ev : e->ev_ring) (void)hipEventDestroy(ev);
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
    // worker-sized batches arrive on the control stream: one-wave blocks (engine_types.hpp launch_finalize_batch)
    const uint32_t latency_blocks = (e->ctl_stream && s == e->ctl_stream) ? 64u : 0u;
    CallScope scope(e, s); // hands the workspace set and the counter block back on every exit path
    int rc = ensure_workspace(e, n, s);
    if (rc != GRV_OK) return rc;
    SegmentParams P = make_segment_params(e, *opt);
    GRV_HIP(e, hipMemsetAsync(e->d_counters, 0, 4 * sizeof(uint32_t), s));
    rc = begin_frame_stats(e, s);
    if (rc != GRV_OK) return rc;
    GRV_HIP(e, launch_init_states(opt->metric_kind, e->ws, P, d_states, opt->initial_step,
                                  opt->method == GRV_METHOD_RKF45, s));
    // independent rays diverge freely in a batch.  Default: one resident launch whose waves
    // refill finished lanes from a device-side cursor (no host round trip); segment_tries > 0
    // asks for the relaunch + live-list compaction schedule instead.
    if (opt->segment_tries > 0) {
        rc = run_segments(e, *opt, P, (uint32_t)opt->segment_tries, s, false);
        if (rc != GRV_OK) return rc;
    } else {
        P.max_tries = opt->segment_tries < 0 ? (uint32_t)(-(int64_t)opt->segment_tries) : kRefillPeriod;
        P.try_cap = try_bound(e, opt->max_steps);
        GRV_HIP(e, launch_refill(opt->arith, opt->metric_kind, opt->method, e->ws, P,
                                 e->d_counters + 2, e->n_cu, s, latency_blocks));
        e->last_launches += 1;
    }
    GRV_HIP(e, launch_finalize_batch(e->ws, d_out_states, d_steps, d_termination, d_drift,
                                     e->d_stats, s, latency_blocks));
    return GRV_OK;
}

// host-pointer batches up to this size take the control stream (latency), larger ones the null stream (throughput:
// a quarter of a million rays at high priority would hold a renderer's frames back instead)
constexpr size_t kLatencyBatchRays = 16384;

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
    if (n <= kLatencyBatchRays) {
        // a worker-sized batch: on the engine's high-priority control stream, synchronised alone -- it must not wait
        // for the frames a renderer has queued on the same device (napi/control_latency.js)
        hipStream_t cs;
        rc = control_stream(e, &cs);
        if (rc != GRV_OK) return rc;
        GRV_HIP(e, hipMemcpyAsync(d_in, states, n * 64, hipMemcpyHostToDevice, cs));
        rc = grv_integrate_batch_device(e, n, d_in, opt, d_out, d_steps, d_term, d_drift, cs);
        if (rc != GRV_OK) return rc;
        GRV_HIP(e, hipMemcpyAsync(out_states, d_out, n * 64, hipMemcpyDeviceToHost, cs));
        if (steps) GRV_HIP(e, hipMemcpyAsync(steps, d_steps, n * 4, hipMemcpyDeviceToHost, cs));
        if (termination) GRV_HIP(e, hipMemcpyAsync(termination, d_term, n, hipMemcpyDeviceToHost, cs));
        if (drift) GRV_HIP(e, hipMemcpyAsync(drift, d_drift, n * 8, hipMemcpyDeviceToHost, cs));
        GRV_HIP(e, hipStreamSynchronize(cs));
        return GRV_OK;
    }
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

// Trajectory.path (geodesic/mod.rs:160): the batch call with the recorded path of every ray.
int grv_integrate_paths_device(grv_engine *e, size_t n, const double *d_states, const GrvOptions *opt,
                               size_t max_points, double *d_paths, uint32_t *d_counts,
                               double *d_out_states, uint32_t *d_steps, uint8_t *d_termination,
                               double *d_drift, void *stream) {
    if (!e) return GRV_ERR_INVALID;
    if (!opt || !options_valid(*opt)) return fail(e, GRV_ERR_INVALID, "invalid GrvOptions");
    if (n == 0) return GRV_OK;
    if (!d_states) return fail(e, GRV_ERR_INVALID, "null states");
    if (!d_counts) return fail(e, GRV_ERR_INVALID, "null counts");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!opt->record_path) { // path: None -- the plain batch call, and every Vec has length 0
        const int rc = grv_integrate_batch_device(e, n, d_states, opt, d_out_states, d_steps, d_termination,
                                                  d_drift, stream);
        if (rc != GRV_OK) return rc;
        GRV_HIP(e, hipMemsetAsync(d_counts, 0, n * sizeof(uint32_t), s));
        return GRV_OK;
    }
    if (n > 0x7FFFFFFFull) return fail(e, GRV_ERR_INVALID, "batch too large");
    if (max_points > 0xFFFFFFFFull) return fail(e, GRV_ERR_INVALID, "max_points too large");
    if (max_points > 0 && !d_paths) return fail(e, GRV_ERR_INVALID, "null paths");
    GRV_HIP(e, hipSetDevice(e->device));
    CallScope scope(e, s);
    int rc = ensure_workspace(e, n, s);
    if (rc != GRV_OK) return rc;
    SegmentParams P = make_segment_params(e, *opt);
    rc = begin_frame_stats(e, s);
    if (rc != GRV_OK) return rc;
    GRV_HIP(e, launch_init_states(opt->metric_kind, e->ws, P, d_states, opt->initial_step,
                                  opt->method == GRV_METHOD_RKF45, s));
    P.max_tries = try_bound(e, opt->max_steps);
    P.final_launch = 1;
    GRV_HIP(e, launch_path(opt->arith, opt->metric_kind, opt->method, e->ws, P, d_states, d_paths, d_counts,
                           (uint32_t)max_points, s));
    e->last_launches += 1;
    GRV_HIP(e, launch_finalize_batch(e->ws, d_out_states, d_steps, d_termination, d_drift, e->d_stats, s));
    return GRV_OK;
}

int grv_integrate_paths(grv_engine *e, size_t n, const double *states, const GrvOptions *opt,
                        size_t max_points, double *out_paths, uint32_t *out_counts, double *out_states,
                        uint32_t *steps, uint8_t *termination, double *drift) {
    if (!e) return GRV_ERR_INVALID;
    if (n == 0) return GRV_OK;
    if (!states || !out_counts) return fail(e, GRV_ERR_INVALID, "null states / counts");
    if (!opt) return fail(e, GRV_ERR_INVALID, "invalid GrvOptions");
    const bool rec = opt->record_path != 0;
    if (rec && max_points > 0 && !out_paths) return fail(e, GRV_ERR_INVALID, "null paths");
    if (max_points > 0xFFFFFFFFull || (rec && max_points && n > ((size_t)1 << 40) / 64 / max_points))
        return fail(e, GRV_ERR_INVALID, "path buffer too large"); // more than 1 TiB of path rows
    GRV_HIP(e, hipSetDevice(e->device));
    const size_t sb = align_up(n * 64, 256), ub = align_up(n * 4, 256), bb = align_up(n, 256),
                 db = align_up(n * 8, 256), pb = rec ? align_up(n * max_points * 64, 256) : 0;
    int rc = ensure_stage(e, 2 * sb + 2 * ub + bb + db + pb);
    if (rc != GRV_OK) return rc;
    char *p = static_cast<char *>(e->stage_mem);
    double *d_in = reinterpret_cast<double *>(p);
    double *d_out = reinterpret_cast<double *>(p + sb);
    uint32_t *d_steps = reinterpret_cast<uint32_t *>(p + 2 * sb);
    uint32_t *d_counts = reinterpret_cast<uint32_t *>(p + 2 * sb + ub);
    uint8_t *d_term = reinterpret_cast<uint8_t *>(p + 2 * sb + 2 * ub);
    double *d_drift = reinterpret_cast<double *>(p + 2 * sb + 2 * ub + bb);
    double *d_paths = reinterpret_cast<double *>(p + 2 * sb + 2 * ub + bb + db);
    GRV_HIP(e, hipMemcpy(d_in, states, n * 64, hipMemcpyHostToDevice));
    rc = grv_integrate_paths_device(e, n, d_in, opt, max_points, d_paths, d_counts, d_out, d_steps, d_term,
                                    d_drift, nullptr);
    if (rc != GRV_OK) return rc;
    GRV_HIP(e, hipDeviceSynchronize());
    GRV_HIP(e, hipMemcpy(out_counts, d_counts, n * 4, hipMemcpyDeviceToHost));
    if (rec && max_points) {
        // a ray's row holds min(count, max_points) points; the rest of the caller's row stays untouched
        size_t kmax = 0;
        bool full = true;
        for (size_t i = 0; i < n; ++i) {
            const size_t k = out_counts[i] < max_points ? out_counts[i] : max_points;
            kmax = k > kmax ? k : kmax;
            full = full && k == max_points;
        }
        if (full) {
            GRV_HIP(e, hipMemcpy(out_paths, d_paths, n * max_points * 64, hipMemcpyDeviceToHost));
        } else if (kmax) {
            // ragged rows: the first kmax points of a block of rays leave the device in ONE strided
            // transfer into pinned staging (<= 16 MiB a block), the rows are cut to length on the host
            // -- not one blocking copy per ray (thousands of rays from integrate_batch({recordPath})).
            // Staging above 4 MiB is handed back when the call ends (a one-off large call must not pin
            // host memory until grv_engine_destroy); grv_engine_host_bytes reports what is held.
            const size_t row = kmax * 64;
            size_t rays_per_block = ((size_t)16 << 20) / row;
            rays_per_block = rays_per_block ? rays_per_block : 1;
            rays_per_block = rays_per_block < n ? rays_per_block : n;
            if (e->path_stage_bytes < rays_per_block * row) {
                if (e->path_stage) (void)hipHostFree(e->path_stage);
                e->path_stage = nullptr;
                e->path_stage_bytes = 0;
                GRV_HIP(e, hipHostMalloc(&e->path_stage, rays_per_block * row, hipHostMallocDefault));
                e->path_stage_bytes = rays_per_block * row;
            }
            const char *stage = static_cast<const char *>(e->path_stage);
            for (size_t i0 = 0; i0 < n; i0 += rays_per_block) {
                const size_t m = n - i0 < rays_per_block ? n - i0 : rays_per_block;
                GRV_HIP(e, hipMemcpy2D(e->path_stage, row, d_paths + i0 * max_points * 8, max_points * 64, row, m,
                                       hipMemcpyDeviceToHost));
                for (size_t i = 0; i < m; ++i) {
                    const size_t k = out_counts[i0 + i] < max_points ? out_counts[i0 + i] : max_points;
                    if (k) std::memcpy(out_paths + (i0 + i) * max_points * 8, stage + i * row, k * 64);
                }
            }
            if (e->path_stage_bytes > ((size_t)4 << 20)) {
                (void)hipHostFree(e->path_stage);
                e->path_stage = nullptr;
                e->path_stage_bytes = 0;
            }
        }
    }
    if (out_states) GRV_HIP(e, hipMemcpy(out_states, d_out, n * 64, hipMemcpyDeviceToHost));
    if (steps) GRV_HIP(e, hipMemcpy(steps, d_steps, n * 4, hipMemcpyDeviceToHost));
    if (termination) GRV_HIP(e, hipMemcpy(termination, d_term, n, hipMemcpyDeviceToHost));
    if (drift) GRV_HIP(e, hipMemcpy(drift, d_drift, n * 8, hipMemcpyDeviceToHost));
    return GRV_OK;
}

// One geodesic = one launch: the state travels in the kernel arguments, the kernel runs
// init + integrate + write-back (single_ray_kernel) and stores the result in pinned host memory,
// then the call's sequence number; the host polls that word instead of synchronising the stream
// (a completion signal costs more than the PCIe write).  No workspace, no staging copies.
size_t grv_integrate_ray_relativistic_ex(grv_engine *e, const double *initial_state, size_t n,
                                         size_t steps, double tolerance, int use_kerr_schild,
                                         double *out, uint32_t *steps_taken, uint8_t *termination,
                                         double *max_drift) {
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
    o.arith = e->ray_arith;
    if (!options_valid(o)) { // cannot fail for the fixed options above; kept so a later edit of them is checked
        fail(e, GRV_ERR_INVALID, "integrate_ray_relativistic: invalid options");
        for (int i = 0; i < 8; ++i) out[i] = std::nan("");
        return 8;
    }
    // no Result in the reference FFI: on any failure hand back NaNs so the caller's finite-guard
    // (src/engine/physics-bridge.ts:174-180) trips; the error text stays on the handle
    auto nan_out = [&](const char *why, hipError_t st) {
        fail(e, GRV_ERR_HIP, "integrate_ray_relativistic: %s: %s", why, hipGetErrorString(st));
        for (int i = 0; i < 8; ++i) out[i] = std::nan("");
        return (size_t)8;
    };
    hipError_t st = hipSetDevice(e->device);
    if (st != hipSuccess) return nan_out("hipSetDevice", st);
    SegmentParams P = make_segment_params(e, o);
    P.try_cap = try_bound(e, o.max_steps);
    SingleRayIn in;
    std::memcpy(in.v, initial_state, sizeof in.v);
    const uint32_t seq = ++e->ray_seq ? e->ray_seq : ++e->ray_seq; // never 0 (the block starts zeroed)
    // the one-ray entry's own stream, created by its first call: the runtime multiplexes streams onto a few
    // hardware queues, and an engine that only renders frames should not take one of them
    if (!e->ray_stream) { // highest priority: one wave must not queue behind the frames a renderer has in flight
        if (create_priority_stream(e, &e->ray_stream) != GRV_OK) {
            for (int i = 0; i < 8; ++i) out[i] = std::nan("");
            return 8;
        }
    }
    st = (o.arith == GRV_ARITH_FAST ? launch_single_ray_fast : launch_single_ray)(
        o.metric_kind, P, in, o.initial_step, e->ray_out, seq, e->ray_stream);
    if (st != hipSuccess) return nan_out("launch", st);
    // poll the sequence word; a ray that runs for seconds falls back to a blocking wait
    const uint32_t *flag = &e->ray_out->seq;
    bool done = false;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 1; !done; ++spin) {
        done = __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq;
        if (!done && (spin & 0x3FFu) == 0u &&
            std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200))
            break;
    }
    if (!done) {
        st = hipStreamSynchronize(e->ray_stream);
        if (st != hipSuccess) return nan_out("hipStreamSynchronize", st);
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) return nan_out("no result", hipErrorUnknown);
    }
    std::memcpy(out, e->ray_out->state, 8 * sizeof(double));
    if (steps_taken) *steps_taken = e->ray_out->steps;
    if (termination) *termination = (uint8_t)e->ray_out->term;
    if (max_drift) *max_drift = e->ray_out->drift;
    return 8;
}

size_t grv_integrate_ray_relativistic(grv_engine *e, const double *initial_state, size_t n,
                                      size_t steps, double tolerance, int use_kerr_schild,
                                      double *out) {
    return grv_integrate_ray_relativistic_ex(e, initial_state, n, steps, tolerance, use_kerr_schild,
                                             out, nullptr, nullptr, nullptr);
}

uint32_t grv_tile_pitch(uint32_t width, uint32_t tile_world) { return tile_pitch(width, tile_world); }

// The tile deal, the ONE implementation of it (hosts above the ABI -- distributed.py, the addon --
// ask, they do not recompute): ids 0 .. pitch * rows - 1, id k -> rank k mod world.
uint32_t grv_tiles_total(uint32_t width, uint32_t height, uint32_t tile_world) {
    return tile_pitch(width, tile_world == 0 ? 1u : tile_world) * ((height + 63u) / 64u);
}
uint32_t grv_max_tiles_per_rank(uint32_t width, uint32_t height, uint32_t tile_world) {
    const uint32_t w = tile_world == 0 ? 1u : tile_world;
    return (grv_tiles_total(width, height, w) + w - 1u) / w;
}
uint32_t grv_tiles_of_rank(uint32_t width, uint32_t height, uint32_t tile_world, uint32_t tile_rank,
                           uint32_t *out_ids, uint32_t capacity) {
    GrvRenderParams q{};
    q.width = width;
    q.height = height;
    q.tile_world = tile_world == 0 ? 1u : tile_world;
    q.tile_rank = tile_rank;
    if (q.tile_rank >= q.tile_world) return 0;
    FrameGeom G;
    frame_geometry(q, G);
    for (uint32_t tl = 0; out_ids && tl < G.n_tiles_local && tl < capacity; ++tl)
        out_ids[tl] = tl * G.tile_world + G.tile_rank;
    return G.n_tiles_local;
}
void grv_tile_origin(uint32_t tile, uint32_t width, uint32_t tile_world, uint32_t *x0, uint32_t *y0) {
    const uint32_t pitch = tile_pitch(width, tile_world == 0 ? 1u : tile_world);
    if (x0) *x0 = (tile % pitch) * 64u;
    if (y0) *y0 = (tile / pitch) * 64u;
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
    p->reserved0 = 0;
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
    p->disk_profile = GRV_DISK_PROFILE_SHORTCUT;
    p->schedule = GRV_SCHEDULE_DEFAULT;
}

int grv_render_frame_device(grv_engine *e, const GrvCamera *cam, const GrvRenderParams *p,
                            const GrvFrameBuffers *out, void *stream) {
    if (!e) return GRV_ERR_INVALID;
    if (!cam || !p || !out) return fail(e, GRV_ERR_INVALID, "null argument");
    if (!options_valid(p->opt)) return fail(e, GRV_ERR_INVALID, "invalid GrvOptions");
    if (p->width == 0 || p->height == 0) return fail(e, GRV_ERR_INVALID, "empty frame");
    if (p->reserved0 != 0) return fail(e, GRV_ERR_INVALID, "reserved field must be 0");
    if (p->schedule > GRV_SCHEDULE_SLOT_ORDER) return fail(e, GRV_ERR_INVALID, "unknown schedule %u", p->schedule);
    if (p->tile_world >= 1 && p->tile_rank >= p->tile_world) return fail(e, GRV_ERR_INVALID, "tile_rank >= tile_world"); // 0 = whole frame
    if (p->disk_profile > GRV_DISK_PROFILE_PAGE_THORNE) return fail(e, GRV_ERR_INVALID, "unknown disk_profile");
    hipStream_t s = static_cast<hipStream_t>(stream);
    GRV_HIP(e, hipSetDevice(e->device));
    const bool profile = p->profile != 0 && e->ev_ok;

    FrameGeom G;
    frame_geometry(*p, G);
    const size_t slots = (size_t)G.n_tiles_local * 4096u;
    if (slots > 0x7FFFFFFFull) return fail(e, GRV_ERR_INVALID, "frame too large for one rank");
    CallScope scope(e, s); // hands the workspace set and the counter block back on every exit path
    int rc = begin_frame_stats(e, s);
    if (rc != GRV_OK) return rc;
    if (slots == 0) return GRV_OK;
    rc = ensure_workspace(e, slots, s);
    if (rc != GRV_OK) return rc;

    SegmentParams P = make_segment_params(e, p->opt);
    P.block_order = p->tile_world > 1 ? 1u : 0u; // a rank's share: long middle rows first (engine_types.hpp)
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
        if (p->disk_profile == GRV_DISK_PROFILE_PAGE_THORNE) {
            rc = ensure_disk_lut(e, s);
            if (rc != GRV_OK) return rc;
        }
    }

    CameraDev cd;
    std::memcpy(cd.pos, cam->position, sizeof cd.pos);
    std::memcpy(cd.inv_view, cam->inv_view, sizeof cd.inv_view);
    std::memcpy(cd.inv_proj, cam->inv_proj, sizeof cd.inv_proj);
    std::memcpy(cd.off, cam->pixel_offset, sizeof cd.off);
    {
        // compute.wgsl.ts:172-176: r0, theta0 = acos(y / r0), phi0 = atan2(z, x)
        const double *c = cam->position;
        cd.r0 = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        double cy = c[1] / cd.r0;
        cy = cy < -1.0 ? -1.0 : (cy > 1.0 ? 1.0 : cy);
        cd.theta0 = strictm::sl_acos(cy);
        cd.phi0 = strictm::sl_atan2(c[2], c[0]);
        cd.st = strictm::sl_sin(cd.theta0);
        cd.ct = strictm::sl_cos(cd.theta0);
        cd.sp = strictm::sl_sin(cd.phi0);
        cd.cp = strictm::sl_cos(cd.phi0);
    }

    // profile: four events around the three kernels of the frame, recorded on `s` and resolved
    // by grv_frame_stats -- nothing here waits for the device
    hipEvent_t *ev4 = nullptr;
    if (profile) {
        rc = ring_events(e, &ev4);
        if (rc != GRV_OK) return rc;
        GRV_HIP(e, hipEventRecord(ev4[0], s));
    }
    // one-launch schedule of a whole frame in one-wave blocks: dispatched longest-first by the previous frame's
    // per-wave tries (costs written by this frame's finalize kernel, sorted behind it for the next frame)
    MarchSched sched{nullptr, nullptr};
    int order_parity = -1;
    const uint32_t *head_order = nullptr;
    // (a rank's share of a split frame too: until the end of round 6 those ran centre-out in four-wave blocks, the choice of
    // a measurement that predates the order -- the 4K frame's eighth 3.87 -> 3.49 ms with one frame in flight, 3.49 -> 3.43 ms
    // with two, profiles/r06_ab_rank_share_order.txt; their first frame runs in slot order, every later one longest-first)
#ifndef GRV_SEG_ORDER_RANK_SHARES
#define GRV_SEG_ORDER_RANK_SHARES 1
#endif
    if (p->schedule == GRV_SCHEDULE_DEFAULT && (P.block_order == 0 || GRV_SEG_ORDER_RANK_SHARES) && slots >= kSegOrderMinRays) {
        const uint32_t geom[4] = {p->width, p->height, G.tile_world, G.tile_rank};
        rc = begin_march_order(e, 2, (uint32_t)(slots / 64u), geom, s, &sched, &order_parity);
        if (rc != GRV_OK) return rc;
        if (p->segment_tries == 0) P.order = sched.order;
        else if (e->march_order[2][order_parity].ranked) head_order = sched.order; // the compacting schedule's head start
    }
    GRV_HIP(e, launch_init_pixels(p->opt.metric_kind, e->ws, P, G, cd, p->opt.initial_step,
                                  p->opt.method == GRV_METHOD_RKF45, s));
    if (profile) GRV_HIP(e, hipEventRecord(ev4[1], s));
    // neighbouring pixels take near-identical step counts (8x8-pixel waves run at >99 %
    // lane efficiency at 4K), so the frame default is one long segment; segment_tries
    // selects the compacting wavefront form
    rc = run_segments(e, p->opt, P, p->segment_tries, s, profile, head_order);
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
    S.disk_profile = p->disk_profile;
    S.pt_rin = isco_prograde(e->mass, e->spin_c); // physics/disk.rs:176-177
    S.pt_rout = 50.0 * e->mass;
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
    if (profile) GRV_HIP(e, hipEventRecord(ev4[2], s));
    GRV_HIP(e, launch_finalize_frame(e->ws, G, S, p->shading, e->d_lut,
                                     p->disk_profile == GRV_DISK_PROFILE_PAGE_THORNE ? e->d_disk_lut : nullptr,
                                     out->rgba,
                                     out->final_state, out->steps, out->termination, out->drift,
                                     e->d_stats, e->n_cu, s, sched.cost));
    if (order_parity >= 0) {
        rc = finish_march_order(e, 2, order_parity, s);
        if (rc != GRV_OK) return rc;
    }
    if (profile) {
        GRV_HIP(e, hipEventRecord(ev4[3], s));
        // (no schedule waits for the host any more: the ring's before-integrate..before-shade interval is the
        // integrate pass for every one of them)
        e->ev_loop.resize(e->ev_frames + 1);
        e->ev_loop[e->ev_frames] = 0;
        e->ev_frames += 1;
    }
    return GRV_OK;
}

int grv_frame_stats(grv_engine *e, void *stream, GrvFrameStats *stats) {
    if (!e || !stats) return GRV_ERR_INVALID;
    hipStream_t s = static_cast<hipStream_t>(stream);
    GRV_HIP(e, hipSetDevice(e->device));
    // the counters belong to calls that may sit on other streams than `s`
    if (e->stats_accum) {
        GRV_HIP(e, hipDeviceSynchronize()); // every call since the reset added to this block
    } else {
        const int b = (int)(e->d_stats - e->stats_blocks);
        if (e->stats_used[b]) GRV_HIP(e, hipStreamWaitEvent(s, e->stats_done[b], 0));
    }
    GRV_HIP(e, hipMemcpyAsync(e->h_stats, e->d_stats, sizeof(FrameStatsDev), hipMemcpyDeviceToHost, s));
    GRV_HIP(e, hipStreamSynchronize(s));
    const int rc = resolve_frame_events(e);
    if (rc != GRV_OK) return rc;
    stats_to_abi(e, *e->h_stats, stats);
    return GRV_OK;
}

size_t grv_engine_device_bytes(const grv_engine *e) {
    if (!e) return 0;
    size_t b = 3 * sizeof(FrameStatsDev) + e->stage_bytes + e->post_bytes;
    for (const auto &W : e->wset) b += W.bytes;
    if (e->d_lut) b += (size_t)e->lut_w * e->lut_h * 4 * sizeof(float);
    if (e->d_disk_lut) b += 4096 + kDiskLutWidth * sizeof(double);
    if (e->d_noise) b += 2u * 256u * 256u + 256u * 256u * sizeof(float); // two R8 planes + the f32 texel plane
    if (e->rt.mem) b += (size_t)3 * e->rt.w * e->rt.h * 4 * sizeof(float);
    return b;
}

void *grv_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void grv_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int grv_last_ray_clocks(const grv_engine *e, uint64_t out3[3]) {
    if (!e || !out3 || !e->ray_out) return GRV_ERR_INVALID;
    out3[0] = e->ray_out->loop_cycles;
    out3[1] = e->ray_out->loop_ticks;
    out3[2] = e->ray_out->tries;
    return GRV_OK;
}

int grv_engine_set_ray_arith(grv_engine *e, int32_t arith) {
    if (!e) return GRV_ERR_INVALID;
    if (arith != GRV_ARITH_STRICT && arith != GRV_ARITH_FAST)
        return fail(e, GRV_ERR_INVALID, "grv_engine_set_ray_arith: invalid arith %d", arith);
    e->ray_arith = arith;
    return GRV_OK;
}

// Verification hooks answer only after grv_test_hooks_unlock(GRV_TEST_HOOKS_KEY): a stray call from
// production code cannot truncate rays or reshuffle exchange buffers.
static std::atomic<bool> g_test_hooks{false};
int grv_test_hooks_unlock(uint32_t key) {
    if (key != GRV_TEST_HOOKS_KEY) return GRV_ERR_INVALID;
    g_test_hooks.store(true);
    return GRV_OK;
}
int grv_test_hooks_unlocked(void) { return g_test_hooks.load() ? 1 : 0; }

int grv_test_set_try_bound(grv_engine *e, uint32_t tries) {
    if (!e) return GRV_ERR_INVALID;
    if (!g_test_hooks.load())
        return fail(e, GRV_ERR_INVALID, "grv_test_set_try_bound: verification hooks are locked (grv_test_hooks_unlock)");
    e->try_bound_override = tries;
    return GRV_OK;
}
uint32_t grv_test_try_bound(const grv_engine *e) { return e ? e->try_bound_override : 0u; }

int grv_engine_profile_shader_frames(grv_engine *e, int enable) {
    if (!e) return GRV_ERR_INVALID;
    e->profile_shader = enable != 0;
    return GRV_OK;
}

size_t grv_engine_host_bytes(const grv_engine *e) {
    if (!e) return 0;
    return e->path_stage_bytes + sizeof(FrameStatsDev) + 64; // pinned: path staging, counter read-back, ray result
}

int grv_stats_accumulate(grv_engine *e, int enable) {
    if (!e) return GRV_ERR_INVALID;
    e->stats_accum = enable != 0;
    return GRV_OK;
}

int grv_engine_synchronize(grv_engine *e) {
    if (!e) return GRV_ERR_INVALID;
    GRV_HIP(e, hipSetDevice(e->device));
    GRV_HIP(e, hipDeviceSynchronize());
    return GRV_OK;
}

int grv_frame_stats_reset(grv_engine *e, void *stream) {
    if (!e) return GRV_ERR_INVALID;
    hipStream_t s = static_cast<hipStream_t>(stream);
    GRV_HIP(e, hipSetDevice(e->device));
    // behind every call that may still be adding to the block (they may sit on other streams)
    for (int b = 0; b < 3; ++b)
        if (e->stats_used[b]) GRV_HIP(e, hipStreamWaitEvent(s, e->stats_done[b], 0));
    for (auto &W : e->wset)
        if (W.used) GRV_HIP(e, hipStreamWaitEvent(s, W.done, 0));
    GRV_HIP(e, hipMemsetAsync(e->d_stats, 0, sizeof(FrameStatsDev), s));
    GRV_HIP(e, hipEventRecord(e->stats_cleared, s)); // later calls on other streams start behind the clear
    e->stats_cleared_rec = true;
    for (float &m : e->last_ms) m = 0.f;
    e->last_launches = 0;
    e->ev_frames = 0;
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
            if (X0 >= G.width) break; // pad column of the tile-id grid: no pixels
            const uint32_t w = (X0 + 64 <= G.width) ? 64u : (G.width - X0);
            std::memcpy(dst + ((size_t)Y * G.width + X0) * bpp,
                        src + ((size_t)tl * 4096u + (size_t)py * 64u) * bpp, (size_t)w * bpp);
        }
    }
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
    const double f = strictm::sl_cos(fovy_rad / 2.0) / strictm::sl_sin(fovy_rad / 2.0); // 1 / tan, specified
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

} // extern "C"
