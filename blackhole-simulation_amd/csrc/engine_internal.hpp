// engine_internal.hpp -- the engine object and the host helpers shared by the translation
// units that implement the C ABI of include/gravitas_abi.h:
//   engine.hip          lifecycle, closed forms, batch / single-ray / frame entry points
//   engine_shaders.hip  f32 shader frames, post chain, the two renderers
//   engine_control.hip  LUTs, disk / shadow helpers, SAB protocol, spacetime read-outs
// One engine == one `PhysicsEngine` (physics-engine/gravitas-wasm/src/lib.rs:42-54) bound to one
// HIP device.  No CPU compute path exists: every integrate / render / LUT entry point launches
// HIP kernels and fails with a status code if the device is missing.
#pragma once

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
#include "engine_types.hpp"
#include "spacetime_viz.hpp"

struct grv_engine {
    int device = 0;
    double mass = 1.0;
    double spin = 0.0;   // as given (lib.rs:44-45)
    double spin_c = 0.0; // clamped copy held by the metrics (kerr.rs:48-63)
    int n_cu = 256;
    uint32_t try_bound_override = 0; // grv_test_set_try_bound (verification hook), 0 = the derived bound
    std::string err;

    // ray workspaces (device).  Up to two sets, so a caller that alternates two streams keeps two
    // frames in flight: the tail of one frame's integrate launch (too few waves left to fill the
    // chip) runs under the head of the next.  A call reuses the set of the previous call when it
    // arrives on the same stream (stream order already serialises them); only a call on a
    // DIFFERENT stream takes -- and on first use allocates -- the other set, so one-stream and
    // host-pointer callers hold one workspace.  Each set carries an event recorded at the end of its
    // last use; the next user's stream waits for it, so a set is never touched by two calls at once
    // whatever streams the caller picks.
    struct WorkSet {
        void *mem = nullptr;
        size_t slots = 0, bytes = 0;
        grvhip::RayWorkspace ws{};
        uint32_t *live[2] = {nullptr, nullptr};
        uint32_t *d_counters = nullptr; // [2] refill cursor, [4..6] live counts of the compacting schedule (rotating)
        hipEvent_t done = nullptr;
        hipStream_t last_stream = nullptr;
        bool used = false;
    } wset[3]; // [2]: calls on the control stream (worker-sized batches) -- never behind a frame's workspace
    int wlast = 0;                     // the set the previous call used
    WorkSet *cur = nullptr;            // the set of the call in progress
    grvhip::RayWorkspace ws{};         // == cur->ws
    uint32_t *live[2] = {nullptr, nullptr};
    uint32_t *d_counters = nullptr;
    // frame counters: two device blocks.  Without grv_stats_accumulate successive calls alternate
    // them (a call clears its block behind the block's previous user, so the clear never lands under
    // another stream's still-running finalize kernel); with it every call adds to the current block.
    // d_stats is the block of the call in progress / the last call: what grv_frame_stats reads.
    grvhip::FrameStatsDev *stats_blocks = nullptr; // [3]: two alternating + one for calls on the control stream
    hipEvent_t stats_done[3] = {nullptr, nullptr, nullptr};
    bool stats_used[3] = {false, false, false};
    hipEvent_t stats_cleared = nullptr; // end of the last grv_frame_stats_reset
    bool stats_cleared_rec = false;
    int stats_turn = 0;
    bool stats_open = false;           // a call holds d_stats (begin_frame_stats .. end_frame_stats)
    grvhip::FrameStatsDev *d_stats = nullptr;
    uint32_t *h_counters = nullptr; // pinned
    // compacting schedule: live rays each launch of the last completed pass started with, written by the launches
    // themselves into pinned host memory (engine.hip run_segments: the next pass's forecast; never waited for)
    uint32_t *compact_fb = nullptr;
    uint32_t compact_fb_n = 0, compact_fb_tries = 0; // shape of the pass the entries describe
    grvhip::FrameStatsDev *h_stats = nullptr; // pinned

    // single-ray entry (grv_integrate_ray_relativistic): its own non-blocking stream and a pinned
    // result block the kernel writes and the host polls
    hipStream_t ray_stream = nullptr;   // (created by the first one-ray call, at the device's highest stream priority)
    // control-plane calls that need the device (LUTs, meshes, fields, the math probes): a stream of their own at the
    // highest priority, synchronised alone -- a worker's parameter change must not wait for the frames a renderer has
    // queued on the same device (measured: 82 ms behind three queued 4K frames on the null stream, napi/control_latency.js)
    hipStream_t ctl_stream = nullptr;
    grvhip::SingleRayOut *ray_out = nullptr; // pinned, host-coherent
    uint32_t ray_seq = 0;
    int ray_arith = GRV_ARITH_STRICT; // contract of the one-ray entry (grv_engine_set_ray_arith)

    // staging buffers for host-pointer entry points
    void *stage_mem = nullptr;
    size_t stage_bytes = 0;
    void *path_stage = nullptr; // pinned: ragged Trajectory.path rows on their way out (grv_integrate_paths)
    size_t path_stage_bytes = 0;

    // cached spectrum LUT (device)
    float *d_lut = nullptr;
    uint32_t lut_w = 0, lut_h = 0;
    double lut_tmax = 0.0;
    // cached Page-Thorne temperature table (device, 512 f32 + scratch), keyed by (mass, spin_c)
    float *d_disk_lut = nullptr;
    double disk_lut_mass = 0.0, disk_lut_spin = 0.0;
    bool disk_lut_valid = false;
    // the tables are generated on whichever stream the frame call that needed them received; every
    // frame stream that reads one is ordered behind its generation through these events
    hipEvent_t lut_ready = nullptr, disk_lut_ready = nullptr;

    // frame bookkeeping.  Counters live on the device (d_stats) and are read by grv_frame_stats
    // only; with stats_accum they are not cleared between frames (grv_stats_accumulate), so a
    // frame loop needs no host round trip at all.  profile=1 frames take four events from a ring
    // (before init | before integrate | before shade | after shade); their elapsed times are
    // resolved when the statistics are read, never inside the frame call.
    bool stats_accum = false;
    uint32_t last_launches = 0;      // integrate launches since the counters were last cleared
    float last_ms[5] = {0, 0, 0, 0, 0}; // init, integrate, compact, shade, total (resolved events)
    hipEvent_t ev[8] = {};           // blocking per-launch timing of the segment-loop schedule
    bool ev_ok = false;
    std::vector<hipEvent_t> ev_ring; // 4 per profiled frame, created on demand
    size_t ev_frames = 0;            // profiled frames whose events are still unresolved
    std::vector<uint8_t> ev_loop;    // per pending frame: integrate was timed launch by launch
    bool profile_shader = false;     // grv_engine_profile_shader_frames: f32 march launches take ring events too

    // renderer layer (grv_webgpu_render / grv_webgl_render): full-size RGBA f32 targets
    struct Targets {
        float *mem = nullptr; // [3][h][w][4]: scene / compute texture, history ping, history pong
        uint32_t w = 0, h = 0;
        uint32_t hist = 0;   // webgpu: currentHistoryIndex; webgl: currentWriteIndex
        uint32_t frames = 0; // frameCount
    } rt;
    void *post_mem = nullptr; // bloom render targets (bright, blur ping/pong)
    size_t post_bytes = 0;
    // renderer-layer calls that arrive on different image streams (engine_images.hip) follow each other
    // through the history targets and the bloom scratch: each waits for the end of the previous one
    hipEvent_t chain_done = nullptr;
    bool chain_rec = false;
    // measured-cost dispatch order of the FAST marches (engine_types.hpp MarchSched): per march kind (0 GLSL,
    // 1 packed WGSL) and frame parity one {cost, order} pair; a frame reads the order its parity's previous
    // frame produced.  `ready` orders a user on another stream behind the sort that wrote the order.
    // (kind 2: the f64 segment kernel's one-launch schedule -- costs are the waves' tries, written by the
    // finalize kernel)
    struct MarchOrder {
        uint32_t *mem = nullptr; // [2][n_blocks]: cost, order
        uint32_t n_blocks = 0;   // allocated entries per array
        uint32_t cur = 0;        // blocks of the frame in flight
        // the exact frame geometry the order is a permutation for: {width, height, tile_world, tile_rank, blocks};
        // anything else starts from the identity again (a permutation of another block count would leave
        // blocks undispatched)
        uint32_t geom[5] = {0, 0, 0, 0, 0};
        bool has_order = false;
        bool ranked = false;     // the order comes from a sort of measured costs (not the identity of a first frame)
        hipEvent_t ready = nullptr;
        bool ready_rec = false;
    } march_order[3][2];
    uint32_t march_frames[3] = {0, 0, 0};
    // the f64 order's sort runs beside the frame loop, not in it (a 130 000-entry counting sort in one workgroup is
    // 2 % of a 4 ms close-up frame): queued on a stream of its own behind the finalize kernel that wrote the costs
    hipStream_t sort_stream = nullptr;
    hipEvent_t sort_from = nullptr;
    hipEvent_t head_from = nullptr, head_done = nullptr; // the compacting schedule's head-start launch on sort_stream
    uint8_t *d_noise = nullptr; // [2][256*256] R planes: u_noiseTex, u_blueNoiseTex
    std::vector<float> disk_lut = std::vector<float>(512, 0.0f); // lut_buffer (lib.rs:50, 65-66)
    std::vector<float> sab;
    float *sab_ext = nullptr; // attach_sab (lib.rs:74)
    grvhip::CameraFilter camera, last_good_camera;
    float *sab_block() { return sab_ext ? sab_ext : sab.data(); }
};

namespace grvhost {

using namespace grvhip;

int fail(grv_engine *e, int code, const char *fmt, ...);

#define GRV_HIP(e, call)                                                                   \
    do {                                                                                   \
        hipError_t _st = (call);                                                           \
        if (_st != hipSuccess)                                                             \
            return fail((e), _st == hipErrorOutOfMemory ? GRV_ERR_OOM : GRV_ERR_HIP,        \
                        "%s failed: %s", #call, hipGetErrorString(_st));                   \
    } while (0)

// closed forms (host scalars; gravitas-core/src/metric/{mod,kerr}.rs)
double event_horizon(double m, double spin);
double isco_prograde(double m, double a_star);
double photon_sphere(double m, double a_star);
double dilation(double m, double a_star, double r);
double g_factor(double r, double mass, double spin, double lambda);

int ensure_workspace(grv_engine *e, size_t slots, hipStream_t s);
int release_workspace(grv_engine *e, hipStream_t s);
int ensure_stage(grv_engine *e, size_t bytes);
// the engine's control stream / a stream of the device's highest priority (created on first use)
int control_stream(grv_engine *e, hipStream_t *out);
int create_priority_stream(grv_engine *e, hipStream_t *out);
int ensure_lut(grv_engine *e, uint32_t w, uint32_t h, double tmax, hipStream_t s);
int ensure_disk_lut(grv_engine *e, hipStream_t s);
size_t align_up(size_t x, size_t a);
SegmentParams make_segment_params(const grv_engine *e, const GrvOptions &o);
bool options_valid(const GrvOptions &o);
// head_order: the previous frame's wave table sorted longest-first (device), or null -- the compacting schedule then
// gives its first n_waves / kCompactHeadShare waves a head start (engine.hip)
int run_segments(grv_engine *e, const GrvOptions &o, SegmentParams P, uint32_t seg_tries, hipStream_t s,
                 bool profile, const uint32_t *head_order = nullptr);
int begin_frame_stats(grv_engine *e, hipStream_t s);
int end_frame_stats(grv_engine *e, hipStream_t s);
// Ends a frame / batch call on every exit path, early error returns included: the workspace set and
// the counter block the call took are handed back with their events recorded on the call's stream, so
// a later call on another stream can never run on them concurrently with kernels this call queued.
struct CallScope {
    grv_engine *e;
    hipStream_t s;
    CallScope(grv_engine *e_, hipStream_t s_) : e(e_), s(s_) {}
    CallScope(const CallScope &) = delete;
    CallScope &operator=(const CallScope &) = delete;
    ~CallScope() {
        (void)release_workspace(e, s);
        (void)end_frame_stats(e, s);
    }
};
// hard bound on the integrator tries of one ray (see run_segments)
uint32_t try_bound(const grv_engine *e, uint64_t max_steps);
int resolve_frame_events(grv_engine *e);
int ring_events(grv_engine *e, hipEvent_t **ev4);
uint32_t tile_pitch(uint32_t width, uint32_t world);
void frame_geometry(const GrvRenderParams &p, FrameGeom &G);
void stats_to_abi(const grv_engine *e, const FrameStatsDev &d, GrvFrameStats *out);

// SAB offsets in f32 elements (lib.rs:36-40)
constexpr size_t kOffControl = 0, kOffCamera = 64, kOffPhysics = 128, kOffTelemetry = 256, kOffLuts = 2048;

// {order, cost} of the next frame of march `kind` (0 GLSL, 1 packed WGSL) with n_blocks blocks on stream s
// (n_blocks == 0: no measured order for this form); finish_march_order queues the sort behind the march
int begin_march_order(grv_engine *e, int kind, uint32_t n_blocks, const uint32_t geom[4], hipStream_t s, grvhip::MarchSched *out, int *parity);
int finish_march_order(grv_engine *e, int kind, int parity, hipStream_t s);

template <typename Launch>
int run_shader_frame(grv_engine *e, uint32_t width, uint32_t height, uint32_t tw, uint32_t tr,
                     uint64_t *total_steps, hipStream_t s, int kind, uint32_t sched_blocks_of_slots(uint32_t, int32_t),
                     int32_t budget, Launch &&launch) {
    if (width == 0 || height == 0) return fail(e, GRV_ERR_INVALID, "empty frame");
    if (tw >= 1 && tr >= tw) return fail(e, GRV_ERR_INVALID, "tile_rank >= tile_world"); // 0 = whole frame
    GRV_HIP(e, hipSetDevice(e->device));
    GrvRenderParams q{};
    q.width = width;
    q.height = height;
    q.tile_world = tw;
    q.tile_rank = tr;
    FrameGeom G;
    frame_geometry(q, G);
    const size_t slots = (size_t)G.n_tiles_local * 4096u;
    if (slots > 0x7FFFFFFFull) return fail(e, GRV_ERR_INVALID, "frame too large for one rank");
    CallScope scope(e, s);
    {
        const int rc = begin_frame_stats(e, s);
        if (rc != GRV_OK) return rc;
    }
    // profiled shader frames: the march launch between two pairs of ring events on its own stream
    // (resolved by grv_frame_stats into integrate_ms / launches, as for the f64 frame)
    hipEvent_t *ev4 = nullptr;
    if (e->profile_shader && e->ev_ok) {
        const int rc = ring_events(e, &ev4);
        if (rc != GRV_OK) return rc;
        GRV_HIP(e, hipEventRecord(ev4[0], s));
        GRV_HIP(e, hipEventRecord(ev4[1], s));
    }
    MarchSched sched{nullptr, nullptr};
    int parity = -1;
    const uint32_t sched_blocks = sched_blocks_of_slots ? sched_blocks_of_slots((uint32_t)slots, budget) : 0u;
    if (sched_blocks) {
        const uint32_t geom[4] = {width, height, tw, tr};
        const int rc = begin_march_order(e, kind, sched_blocks, geom, s, &sched, &parity);
        if (rc != GRV_OK) return rc;
    }
    GRV_HIP(e, launch(G, (uint32_t)slots, e->d_stats->steps_part, sched));
    if (parity >= 0) {
        const int rc = finish_march_order(e, kind, parity, s);
        if (rc != GRV_OK) return rc;
    }
    if (ev4) {
        GRV_HIP(e, hipEventRecord(ev4[2], s));
        GRV_HIP(e, hipEventRecord(ev4[3], s));
        e->ev_loop.resize(e->ev_frames + 1);
        e->ev_loop[e->ev_frames] = 0;
        e->ev_frames += 1;
        e->last_launches += 1;
    }
    if (total_steps) {
        GRV_HIP(e, hipMemcpyAsync(e->h_stats, e->d_stats, sizeof(FrameStatsDev), hipMemcpyDeviceToHost, s));
        GRV_HIP(e, hipStreamSynchronize(s));
        *total_steps = stats_total_steps(*e->h_stats);
    }
    return GRV_OK;
}

} // namespace grvhost
