/*
 * gravitas_abi.h -- C ABI of the MI355X geodesic engine (libgravitas_hip.so).
 *
 * Drop-in boundary for the reference's physics FFI:
 *   physics-engine/gravitas-wasm/src/lib.rs:56-465  (#[wasm_bindgen] impl PhysicsEngine)
 * Every entry point names the reference interface it replaces.  Plain C types
 * only: opaque handle, int status codes, caller-allocated outputs, no
 * exceptions across the boundary.  A handle is not thread-safe; distinct
 * handles are independent (same contract as one PhysicsEngine per JS realm,
 * src/workers/physics.worker.ts:35-105).
 *
 * All compute entry points run on the GPU.  There is no CPU fallback: without
 * a usable HIP device grv_engine_create fails with GRV_ERR_NO_DEVICE.
 */
#ifndef GRAVITAS_ABI_H
#define GRAVITAS_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRV_ABI_VERSION 8

typedef struct grv_engine grv_engine;

/* status codes */
enum {
    GRV_OK = 0,
    GRV_ERR_INVALID = 1,   /* bad argument */
    GRV_ERR_NO_DEVICE = 2, /* no HIP device / device index out of range */
    GRV_ERR_HIP = 3,       /* HIP runtime error, see grv_last_error */
    GRV_ERR_OOM = 4
};

/* gravitas-core/src/geodesic/termination.rs:4-17 (#[repr(C)] TerminationReason) */
enum {
    GRV_TERM_NONE = 0,
    GRV_TERM_HORIZON = 1,
    GRV_TERM_ESCAPE = 2,
    GRV_TERM_MAXSTEPS = 3,
    GRV_TERM_DISK_CROSSING = 4
};

/* gravitas-core/src/metric/kerr.rs:17-22 CoordinateSystem + metric/schwarzschild.rs */
enum { GRV_METRIC_KERR_BL = 0, GRV_METRIC_KERR_KS = 1, GRV_METRIC_SCHWARZSCHILD = 2 };

/* gravitas-core/src/geodesic/integrator.rs:13-21 IntegrationMethod */
enum { GRV_METHOD_RKF45 = 0, GRV_METHOD_RK4 = 1, GRV_METHOD_SYMPLECTIC = 2 };

/* arithmetic contract of the RKF45 kernel:
 *  STRICT: reference operation order, IEEE divide/sqrt, no FMA contraction;
 *  FAST  : algebraically identical right-hand side with shared reciprocals and
 *          FMA contraction (differs from STRICT by rounding only). */
enum { GRV_ARITH_STRICT = 0, GRV_ARITH_FAST = 1,
       /* WGSL march only: the FAST contract with two rays per lane on the packed-f32 VALU ops */
       GRV_ARITH_FAST_PACKED = 2 };

/* gravitas-core/src/geodesic/integrator.rs:24-33 IntegrationOptions
 * (+ step_size carried by IntegrationMethod::{RK4,Symplectic}) */
typedef struct {
    int32_t method;
    int32_t metric_kind;
    double tolerance;
    double initial_step;
    uint64_t max_steps;
    double escape_radius;
    uint64_t renormalize_interval; /* 0 = never (Rust would panic) */
    double step_size;
    int32_t arith;                 /* GRV_ARITH_* */
    int32_t segment_tries;         /* batch calls, how finished rays give up their lanes:
                                      0  = engine default: one resident launch, waves refill
                                           finished lanes from a device-side cursor every 8 tries;
                                      <0 = the same with a refill check every -segment_tries tries;
                                      >0 = relaunch with live-ray compaction every segment_tries
                                           tries (the next launch reads the live count from device
                                           memory: the whole pass is queued without a host wait).
                                      Results do not depend on this field. */
    int32_t record_path;           /* integrator.rs:32 IntegrationOptions.record_path: read by
                                      grv_integrate_paths only (the other entry points have no path
                                      output, as a Trajectory with path: None) */
    int32_t reserved;              /* must be 0 */
} GrvOptions;

/* f64 mirror of the CameraUniforms fields the compute kernel reads
 * (src/shaders/types.wgsl.ts:6-17; src/types/webgpu.ts:95-116).  Matrices are
 * column-major (gl-matrix / WGSL). */
typedef struct {
    double position[3];
    double inv_view[16];
    double inv_proj[16];
    double pixel_offset[2]; /* uv = (id + offset)/size; (0.5,0.5) = pixel centres;
                               WGSL frame-0 Halton jitter = (0, -1/6) */
} GrvCamera;

typedef struct {
    uint32_t width, height;
    GrvOptions opt;
    int32_t shading;      /* 0 endpoints only, 1 thin-disk (T x g) LUT shading */
    int32_t reserved0;    /* must be 0 (the f32 marches have their own entry points below) */
    double disk_inner;    /* <= 0: prograde ISCO */
    double disk_outer;    /* src/shaders/compute.wgsl.ts:217 -> 30 M */
    double disk_temp;     /* K */
    double disk_opacity;  /* alpha per plane crossing */
    double exposure;
    uint32_t lut_width, lut_height;
    double lut_max_temp;
    /* image-plane partition for multi-GPU: this call renders the 64x64-pixel
     * tiles k with k % tile_world == tile_rank, packed in tile order
     * (physics-engine/_legacy_src/tiling.rs:38-56 row-major grid). 1/0 = whole frame.
     * Tile (tx, ty) has id ty * P + tx, P = grv_tile_pitch(width, tile_world): the first integer
     * >= ceil(width / 64) coprime with tile_world, so the deal shifts from row to row instead of
     * handing a rank whole tile columns; ids in the pad columns hold no pixels. */
    uint32_t tile_world, tile_rank;
    uint32_t segment_tries; /* 0 = engine default: ONE integrate launch that runs every ray to its
                               end, queued on the stream without any host wait;
                               K > 0: the compacting wavefront schedule, K tries per launch, live
                               rays re-listed between launches; every launch after the first takes
                               the list's length from device memory, so this pass too is queued at
                               once (launch count and grids follow the live counts the previous
                               pass's launches reported; a last launch runs what is left to its end).
                               Results do not depend on this field. */
    uint32_t profile;       /* 1: record HIP events around the frame's kernels on the stream; their
                               elapsed times are resolved by grv_frame_stats (GrvFrameStats.*_ms),
                               the frame call itself never waits */
    uint32_t disk_profile;  /* GRV_DISK_PROFILE_*: radial temperature profile of the disk shading */
    uint32_t schedule;      /* GRV_SCHEDULE_*: how the integrate launch(es) are ordered.  Results do not depend on it. */
} GrvRenderParams;

/* GRV_SCHEDULE_DEFAULT: the engine's choice -- with segment_tries == 0, one launch whose waves start
 *   longest-first by the PREVIOUS frame's per-wave try counts (a renderer's frames resemble their
 *   predecessors; whole frames of >= 65 536 rays; the first frame of a geometry runs in slot order):
 *   the few waves that graze the photon ring no longer start in the middle of the launch and keep the
 *   chip waiting at its end;
 * GRV_SCHEDULE_SLOT_ORDER: waves start in slot (tile) order, as every launch did before ABI 8. */
enum { GRV_SCHEDULE_DEFAULT = 0, GRV_SCHEDULE_SLOT_ORDER = 1 };

/* radial temperature profile T(r)/T_max of the thin-disk shading:
 *  SHORTCUT    : (isco/r)^3/4 (1 - sqrt(isco/r))^1/4, the shader's closed form
 *                (src/shaders/blackhole/chunks/disk.ts:100-102);
 *  PAGE_THORNE : the 512-entry Novikov-Thorne / Page-Thorne table of generate_disk_lut
 *                (gravitas-core/src/physics/disk.rs:175-201: entry i at r = isco + (i/511)(50M - isco),
 *                normalised to its maximum), sampled with linear interpolation as the uploaded
 *                u_diskLUT texture is (src/rendering/webgl/renderer.ts LINEAR, CLAMP_TO_EDGE). */
enum { GRV_DISK_PROFILE_SHORTCUT = 0, GRV_DISK_PROFILE_PAGE_THORNE = 1 };

typedef struct {
    uint64_t rays;
    uint64_t accepted_steps;
    uint64_t rkf_tries;
    uint64_t term_count[5];
    uint64_t crossings;
    double max_drift;
    uint32_t launches;      /* integrate-kernel launches (segments) */
    float init_ms, integrate_ms, compact_ms, shade_ms, total_ms; /* HIP-event times, profile=1 */
} GrvFrameStats; /* with grv_stats_accumulate(e, 1): sums (max for max_drift) over every frame since
                    the last grv_frame_stats_reset */

/* Device-resident outputs of one frame (all optional except none: pass NULL to skip).
 * Pixel order: row-major over the rendered tile set (whole frame when tile_world<=1). */
typedef struct {
    float *rgba;           /* [n][4] */
    double *final_state;   /* [n][8] AoS GeodesicState (geodesic/mod.rs:23-30) */
    uint32_t *steps;       /* [n] accepted steps */
    uint8_t *termination;  /* [n] */
    double *drift;         /* [n] max |H| */
} GrvFrameBuffers;

/* ---- lifecycle: `new PhysicsEngine(mass, spin)` lib.rs:59 ; update_params lib.rs:78 ---- */
int grv_engine_create(double mass, double spin, int device, grv_engine **out);
/* wasm-bindgen's .free(): waits for everything queued on the handle's device (frames may still be running on a caller's
 * or a device image's stream) before the workspaces go; images created through the handle stay valid */
void grv_engine_destroy(grv_engine *e);
const char *grv_last_error(const grv_engine *e);
int grv_abi_version(void);
int grv_update_params(grv_engine *e, double mass, double spin);

/* ---- closed forms: lib.rs:85-105, 202-205 ---- */
double grv_compute_horizon(const grv_engine *e);
double grv_compute_isco(const grv_engine *e);
double grv_compute_photon_sphere(const grv_engine *e);
double grv_compute_dilation(const grv_engine *e, double r);
double grv_compute_g_factor(const grv_engine *e, double r, double lambda);

/* ---- the path entry: integrate_ray_relativistic lib.rs:422-464
 * (== integratePhotonGeodesic in BASELINE.json).  n < 8 echoes the input
 * (lib.rs:429-431).  Returns the number of doubles written to out. */
size_t grv_integrate_ray_relativistic(grv_engine *e, const double *initial_state, size_t n,
                                      size_t steps, double tolerance, int use_kerr_schild,
                                      double *out);
/* the same call with the Trajectory scalars of geodesic/mod.rs:150-161 (any may be NULL) */
size_t grv_integrate_ray_relativistic_ex(grv_engine *e, const double *initial_state, size_t n,
                                         size_t steps, double tolerance, int use_kerr_schild,
                                         double *out, uint32_t *steps_taken, uint8_t *termination,
                                         double *max_drift);

/* Arithmetic contract of grv_integrate_ray_relativistic[_ex] on this handle (an extension: gravitas-wasm
 * has one arithmetic).  GRV_ARITH_STRICT, the default: the reference's operation order, the oracle's bits.
 * GRV_ARITH_FAST: the same equations with shared reciprocals and FMA -- rounding differences only (<= 1e-5
 * relative on the end state, median <= 1e-9: tests/test_full_frame_parity.py) at a third of the
 * instructions, which for one ray on one wave is a third of the time. */
int grv_engine_set_ray_arith(grv_engine *e, int32_t arith);

/* the device's own clocks around the try loop of the LAST grv_integrate_ray_relativistic* call:
 * out3 = {shader cycles (s_memtime), ticks of the constant 100 MHz counter (s_memrealtime), integrator
 * tries}.  What one step of the serial chain costs without the launch and the PCIe round trip. */
int grv_last_ray_clocks(const grv_engine *e, uint64_t out3[3]);

/* ---- batch extension: n independent integrate() calls (geodesic/mod.rs:180-253).
 * Host pointers; states AoS [n][8].  steps/termination/drift may be NULL. */
int grv_integrate_batch(grv_engine *e, size_t n, const double *states, const GrvOptions *opt,
                        double *out_states, uint32_t *steps, uint8_t *termination,
                        double *drift);
/* same with device pointers, asynchronous on `stream` (a hipStream_t, NULL = default) */
int grv_integrate_batch_device(grv_engine *e, size_t n, const double *d_states,
                               const GrvOptions *opt, double *d_out_states, uint32_t *d_steps,
                               uint8_t *d_termination, double *d_drift, void *stream);

/* ---- Trajectory.path: geodesic/mod.rs:150-161 (the visualised geodesics) ----
 * n independent integrate() calls that also return what the reference collects under
 * IntegrationOptions.record_path (integrator.rs:32): point 0 is the initial state as handed in,
 * pushed BEFORE the initial renormalize_null (mod.rs:193-197, 200), then the state after every
 * completed loop body (mod.rs:241-243) -- 1 + steps_taken points.
 *   out_paths  [n][max_points][8]  AoS GeodesicState rows (the Vec<GeodesicState> of each ray);
 *   out_counts [n]                 length of the reference's Vec: 1 + steps_taken, or 0 when
 *                                  opt->record_path == 0 (path: None; out_paths is not touched).
 * Only the first min(count, max_points) points of a ray are stored; the rest of its row is not
 * written.  out_states / steps / termination / drift as grv_integrate_batch (any may be NULL).
 * One launch runs every ray to its end: meant for the few rays a scene draws, not for frames. */
int grv_integrate_paths(grv_engine *e, size_t n, const double *states, const GrvOptions *opt,
                        size_t max_points, double *out_paths, uint32_t *out_counts,
                        double *out_states, uint32_t *steps, uint8_t *termination, double *drift);
/* same with device pointers, asynchronous on `stream` */
int grv_integrate_paths_device(grv_engine *e, size_t n, const double *d_states, const GrvOptions *opt,
                               size_t max_points, double *d_paths, uint32_t *d_counts,
                               double *d_out_states, uint32_t *d_steps, uint8_t *d_termination,
                               double *d_drift, void *stream);

/* ---- frame: pixel->state of src/shaders/compute.wgsl.ts:159-187 + integrate + shading ---- */
size_t grv_frame_ray_count(const GrvRenderParams *p); /* rays this rank renders */
uint32_t grv_tile_pitch(uint32_t width, uint32_t tile_world);
/* The tile deal (physics-engine/_legacy_src/tiling.rs:38-56: row-major 64x64 tile grid; here on the
 * pitch above and dealt round-robin, id k -> rank k mod tile_world).  Hosts above the ABI ask these
 * instead of recomputing the deal (blackhole-simulation_amd/distributed.py is thin calls into them):
 *   grv_tiles_total        ids of the frame, pad column(s) included: pitch * ceil(height / 64)
 *   grv_max_tiles_per_rank size of a rank's send buffer / receive slot in tiles
 *   grv_tiles_of_rank      the ids `tile_rank` renders, in its packed order; writes at most `capacity`
 *                          of them to out_ids (may be NULL) and returns how many there are
 *   grv_tile_origin        pixel (x0, y0) of an id; x0 >= width for an id in a pad column */
uint32_t grv_tiles_total(uint32_t width, uint32_t height, uint32_t tile_world);
uint32_t grv_max_tiles_per_rank(uint32_t width, uint32_t height, uint32_t tile_world);
uint32_t grv_tiles_of_rank(uint32_t width, uint32_t height, uint32_t tile_world, uint32_t tile_rank,
                           uint32_t *out_ids, uint32_t capacity);
void grv_tile_origin(uint32_t tile, uint32_t width, uint32_t tile_world, uint32_t *x0, uint32_t *y0);
int grv_render_frame(grv_engine *e, const GrvCamera *cam, const GrvRenderParams *p,
                     float *rgba_host, GrvFrameStats *stats);
int grv_render_frame_device(grv_engine *e, const GrvCamera *cam, const GrvRenderParams *p,
                            const GrvFrameBuffers *out, void *stream);
/* synchronises `stream` and reads the counters of the last frame (the only host wait of the
 * device-pointer frame path: grv_render_frame_device itself returns with its kernels queued) */
int grv_frame_stats(grv_engine *e, void *stream, GrvFrameStats *stats);
/* enable != 0: frame / batch / shader-frame calls stop clearing the device-side counters, so a
 * loop of frames accumulates them in HBM and one grv_frame_stats after the loop reads the sums */
int grv_stats_accumulate(grv_engine *e, int enable);
/* waits for everything queued on the engine's device (every stream, every image) */
int grv_engine_synchronize(grv_engine *e);
int grv_frame_stats_reset(grv_engine *e, void *stream);
/* device memory the handle holds right now (ray workspaces, tables, staging, render targets), bytes.
 * A caller that keeps every frame / batch call on ONE stream holds one ray workspace (~170 B per ray
 * slot); the second one exists only once calls have arrived on two different streams. */
size_t grv_engine_device_bytes(const grv_engine *e);
/* page-locked HOST memory the handle holds (ragged-path staging of grv_integrate_paths -- at most
 * 4 MiB of it survives a call --, the counter read-back block, the one-ray result block), bytes */
size_t grv_engine_host_bytes(const grv_engine *e);
/* enable != 0: grv_render_frame_wgsl / _glsl bracket their march launch with HIP events on the
 * launch stream (resolved by grv_frame_stats into integrate_ms / launches, as profile = 1 does for
 * the f64 frame).  Off by default: an unprofiled frame records nothing. */
int grv_engine_profile_shader_frames(grv_engine *e, int enable);
/* Page-locked host memory for the host-pointer entry points (grv_render_frame, grv_integrate_batch,
 * grv_render_frame_multi, *_render_host): a frame rendered into it leaves the device in one DMA at
 * the full PCIe rate instead of going through the runtime's pageable staging.  The N-API addon hands
 * it to JS as an external ArrayBuffer (renderFrame({out})).  NULL on failure. */
void *grv_host_alloc(size_t bytes);
void grv_host_free(void *p);
/* host-only: scatter packed tile-order pixels of `rank` into a row-major W x H x C image */
int grv_unpack_tiles(const GrvRenderParams *p, uint32_t rank, const void *packed, void *image,
                     size_t bytes_per_pixel);

/* device version (rank-0 side of the RCCL gather); bytes_per_pixel must be a multiple of 4 */
int grv_unpack_tiles_device(grv_engine *e, const GrvRenderParams *p, uint32_t rank,
                            const void *d_packed, void *d_image, size_t bytes_per_pixel,
                            void *stream);

/* ---- the image plane across the GPUs of one node (SURVEY 8(b) device_mask, 8(e)) ----
 * One handle = G ranks, one per device of `device_mask` (bit d = HIP device d; rank order =
 * ascending device index, rank 0 assembles), each with its own engine, host thread and two streams.
 * A frame is cut in 64x64 tiles dealt round-robin (tile k -> rank k mod G, the row-major grid of
 * physics-engine/_legacy_src/tiling.rs:38-56 on the pitch of grv_tile_pitch); every rank renders
 * its share, ONE gather brings the finished tiles to rank 0, rank 0 de-interleaves them into the
 * caller's row-major image.  The calls queue their work and return; the image is complete in the
 * order of `root_stream` (a hipStream_t of rank 0's device, NULL = default).  Successive frames
 * alternate two sets of streams and buffers: two frames are in flight. */
typedef struct grv_multi grv_multi;
enum { GRV_TRANSPORT_AUTO = 0,      /* RCCL for G > 1 real devices, else peer copy */
       GRV_TRANSPORT_RCCL = 1,      /* one ncclGroup of G-1 ncclSend / ncclRecv pairs over xGMI
                                       (librccl is bound with dlopen when a handle asks for it) */
       GRV_TRANSPORT_PEER_COPY = 2  /* each rank pushes its tiles with hipMemcpyPeerAsync */ };
int grv_engine_create_multi(double mass, double spin, uint64_t device_mask, int transport,
                            grv_multi **out);
/* G virtual ranks on ONE device (peer-copy transport): the whole assembly path on a one-GPU box */
int grv_engine_create_multi_virtual(double mass, double spin, int device, int ranks, grv_multi **out);
void grv_multi_destroy(grv_multi *m);
const char *grv_multi_last_error(const grv_multi *m);
int grv_multi_rank_count(const grv_multi *m);
int grv_multi_rank_device(const grv_multi *m, int rank);
int grv_multi_transport(const grv_multi *m);
/* why the calling thread's last grv_engine_create_multi* failed ("" after a success): the RCCL /
 * HIP error text, also written to stderr.  A handle that asks for RCCL and cannot have it (library
 * absent, ncclCommInitAll failing) is REFUSED -- the transport never degrades silently. */
const char *grv_multi_create_error(void);
/* binds librccl (dlopen; the environment variable GRV_RCCL_LIBRARY names a non-standard copy and is
 * then the only candidate) and reports ncclGetVersion's code (e.g. 22703).  No device needed.
 * GRV_ERR_NO_DEVICE with the loader's message in `msg` when the library or one of its entry points
 * is missing; the attempt leaves nothing half-bound. */
int grv_rccl_probe(int *version, char *msg, size_t msg_len);
/* one rank's counters and event times (grv_multi_frame_stats sums / maxes them): lets a scaling
 * point be audited rank by rank */
int grv_multi_rank_frame_stats(grv_multi *m, int rank, GrvFrameStats *stats);
grv_engine *grv_multi_engine(grv_multi *m, int rank); /* rank's engine: closed forms, LUT entry points ... */
int grv_multi_update_params(grv_multi *m, double mass, double spin);
/* p->tile_world must be 0 or 1 (the handle deals the tiles).  d_rgba: W x H x 4 f32 on rank 0's device */
int grv_render_frame_multi_device(grv_multi *m, const GrvCamera *cam, const GrvRenderParams *p,
                                  float *d_rgba, void *root_stream);
/* same into host memory (one D2H copy of the assembled image), optional summed statistics */
int grv_render_frame_multi(grv_multi *m, const GrvCamera *cam, const GrvRenderParams *p,
                           float *rgba_host, GrvFrameStats *stats);
/* What the ONE exchange per frame carries.  RGBA32F (default): the ranks' f32 pixels, 16 B each.
 * RGBA16F: four binary16 channels, 8 B per pixel -- the reference's own compute-pass output format
 * (rgba16float storage texture, src/rendering/webgpu/renderer.ts:163-176; written by
 * src/shaders/compute.wgsl.ts:147-258): every rank narrows its share (round to nearest even) before
 * it travels, rank 0 widens it into the caller's f32 image while de-interleaving the tiles.  The
 * assembled image then equals the one-device frame rounded through binary16 channel by channel,
 * for any G (G = 1 included), bit for bit.  Changing the format waits for the handle's queued frames.
 * grv_multi_exchange_bytes_per_frame: bytes that cross between devices for a width x height frame. */
enum { GRV_EXCHANGE_RGBA32F = 0, GRV_EXCHANGE_RGBA16F = 1 };
int grv_multi_set_exchange_format(grv_multi *m, int format);
int grv_multi_exchange_format(const grv_multi *m);
size_t grv_multi_exchange_bytes_per_frame(const grv_multi *m, uint32_t width, uint32_t height);
int grv_multi_synchronize(grv_multi *m);
int grv_multi_stats_accumulate(grv_multi *m, int enable);
int grv_multi_frame_stats_reset(grv_multi *m);
/* waits for every rank, then sums the ranks' counters (max for max_drift and the event times) */
int grv_multi_frame_stats(grv_multi *m, GrvFrameStats *stats);

/* ---- verification hooks: explicit calls on a handle (nothing is read from the environment) ----
 * grv_test_set_try_bound: hard bound on the integrator tries of one ray (0 restores the derived
 * 160 max_steps + 64, which a correct kernel never reaches); a tiny value makes the "a ray can
 * never hang the device" exit reachable: such rays end as GRV_TERM_MAXSTEPS.
 * grv_multi_test_self_exchange: rank 0's own share travels through the transport as well (send
 * buffer -> exchange -> receive slot) instead of being rendered into its slot, so that every
 * transport call runs on a one-device box.  Waits for the handle's queued frames.
 *
 * Both hooks are LOCKED in a freshly loaded library and return GRV_ERR_INVALID until the process has
 * called grv_test_hooks_unlock(GRV_TEST_HOOKS_KEY) (tests do; product hosts never): a stray call
 * cannot truncate rays or reshuffle exchange buffers.  grv_test_try_bound reads the override back
 * (0 = none), so a result produced under a hook can be told from a production one. */
#define GRV_TEST_HOOKS_KEY 0x47525654u /* "GRVT" */
int grv_test_hooks_unlock(uint32_t key);
int grv_test_hooks_unlocked(void);
int grv_test_set_try_bound(grv_engine *e, uint32_t tries);
uint32_t grv_test_try_bound(const grv_engine *e);
int grv_multi_test_self_exchange(grv_multi *m, int enable);
/* grv_multi_test_inject_fault (locked like the hooks above): the NEXT frame of the handle fails where `kind`
 * says, on rank `rank` -- GRV_FAULT_RENDER: the rank's render call; GRV_FAULT_PEER_COPY: its push to rank 0
 * (peer-copy transport); GRV_FAULT_SEND: its ncclSend inside the group (RCCL transport).  The frame call then
 * returns GRV_ERR_HIP with the rank and the cause in grv_multi_last_error, no group is left open, and the
 * handle renders the following frame as if nothing had happened (tests/test_gpu_multi_native.py). */
enum { GRV_FAULT_NONE = 0, GRV_FAULT_RENDER = 1, GRV_FAULT_SEND = 2, GRV_FAULT_PEER_COPY = 3 };
int grv_multi_test_inject_fault(grv_multi *m, int kind, int rank);

/* ---- f32 march loops of the reference's GPU shaders ----
 * Uniform blocks as the shaders receive them.  Outputs are device pointers: RGBA f32
 * [n][4] and (optional) per-pixel step counts, in this rank's pixel order. */
typedef struct { /* src/shaders/compute.wgsl.ts + types.wgsl.ts:6-30 */
    uint32_t width, height;
    float inv_view[16], inv_proj[16], position[3];
    float mass, spin;        /* PhysicsParams.mass / .spin */
    float jitter[2];         /* halton(frame)-0.5 in pixels (compute.wgsl.ts:154-157); 0 = none */
    int32_t max_steps;       /* override MAX_STEPS, default 150 (compute.wgsl.ts:13) */
    uint32_t tile_world, tile_rank;
    int32_t arith;           /* GRV_ARITH_STRICT: the shader's operation order;
                                GRV_ARITH_FAST: same equations, shared reciprocal + FMA (f32 rounding only) */
    int32_t stars;           /* 1 (default): the escape-branch star hash, compute.wgsl.ts:199-206 */
} GrvWgslParams;

/* ShaderManager's #defines (src/shaders/manager.ts:61-82) as GrvGlslParams.features bits */
#define GRV_GLSL_LENSING 1u     /* ENABLE_LENSING */
#define GRV_GLSL_DISK 2u        /* ENABLE_DISK */
#define GRV_GLSL_DOPPLER 4u     /* ENABLE_DOPPLER */
#define GRV_GLSL_STARS 8u       /* ENABLE_STARS */
#define GRV_GLSL_PHOTON_GLOW 16u /* ENABLE_PHOTON_GLOW */
#define GRV_GLSL_JETS 32u       /* ENABLE_JETS (only with DISK, manager.ts:72-73) */
#define GRV_GLSL_REDSHIFT 64u   /* ENABLE_REDSHIFT */
#define GRV_GLSL_DITHER 128u    /* blue-noise start offset, fragment.glsl.ts:104-108 (always on upstream) */
/* the reference's default preset "high-quality" (src/configs/simulation.config.ts:48-60, 77-78) */
#define GRV_GLSL_FEATURES_DEFAULT \
    (GRV_GLSL_LENSING | GRV_GLSL_DISK | GRV_GLSL_DOPPLER | GRV_GLSL_STARS | GRV_GLSL_PHOTON_GLOW | \
     GRV_GLSL_JETS | GRV_GLSL_DITHER)

typedef struct { /* src/shaders/blackhole/chunks/common.ts:8-35 uniforms */
    uint32_t width, height;
    float mass;              /* u_mass */
    float spin;              /* u_spin as uploaded: spin * mass (src/rendering/webgl/renderer.ts:326) */
    float zoom;              /* u_zoom = zoom * 2 (renderer.ts:327) */
    float mouse[2];          /* u_mouse */
    float disk_size, disk_scale_height, disk_density, disk_temp, lensing_strength, time;
    float turbulence;        /* >= 0: stands in for noise()*0.5 + noise()*0.25 of disk.ts:55;
                                < 0: the two fetches from the engine's noise texture */
    int32_t max_ray_steps;   /* u_maxRaySteps (shader clamps to 500, fragment.glsl.ts:115) */
    int32_t tone_map;        /* 0: ENABLE_LINEAR_OUTPUT, 1: ACES + gamma (fragment.glsl.ts:327-331) */
    uint32_t tile_world, tile_rank;
    uint32_t features;       /* GRV_GLSL_* */
    int32_t quality;         /* 0: RAY_QUALITY_OFF/LOW indicator path (fragment.glsl.ts:76-88), else march */
    float show_redshift;     /* u_show_redshift */
    float show_kerr_shadow;  /* u_show_kerr_shadow */
    float debug;             /* u_debug */
    float cam_pos[3];        /* u_camPos; |.| <= 0.001 selects the mouse camera (renderer.ts:312-316) */
    float cam_quat[4];       /* u_camQuat xyzw */
    float shadow_count;      /* u_shadowCount */
    float shadow_curve[64][2]; /* u_shadowCurve (alpha, beta) from compute_shadow_curve */
    int32_t arith;           /* GRV_ARITH_STRICT: the shader's operation order, IEEE divide/sqrt;
                                GRV_ARITH_FAST: FMA + reciprocal-based divide/sqrt + polynomial
                                sin/cos (f32 rounding differences only) */
} GrvGlslParams;

/* The shader's two 256x256 RGBA8 textures (u_noiseTex LINEAR/REPEAT, u_blueNoiseTex
 * NEAREST/REPEAT; only .r is sampled).  Upstream fills them with Math.random()
 * (src/utils/webgl-utils.ts:259-305, not reproducible); an engine starts with seeded ones
 * (grv_seeded_noise_rgba8, seeds 1 and 2).  Pass the `data` arrays createNoiseTexture /
 * createBlueNoiseTexture would upload to override; NULL keeps the current plane. */
void grv_seeded_noise_rgba8(uint32_t seed, uint32_t size, uint8_t *rgba);
int grv_set_glsl_noise(grv_engine *e, const uint8_t *noise_rgba8_256, const uint8_t *blue_rgba8_256);

void grv_wgsl_params_default(uint32_t width, uint32_t height, const GrvCamera *cam, double mass,
                             double spin, GrvWgslParams *p);
void grv_glsl_params_default(uint32_t width, uint32_t height, double mass, double spin,
                             GrvGlslParams *p);
/* accepted steps of the last shader frame are returned through *total_steps (host) after a
 * stream synchronise; d_steps may be NULL */
int grv_render_frame_wgsl(grv_engine *e, const GrvWgslParams *p, float *d_rgba, uint32_t *d_steps,
                          uint64_t *total_steps, void *stream);
int grv_render_frame_glsl(grv_engine *e, const GrvGlslParams *p, float *d_rgba, uint32_t *d_steps,
                          uint64_t *total_steps, void *stream);
/* the WGSL compute march over the ranks of a multi-GPU handle (BASELINE configs[3]); as
 * grv_render_frame_multi_device.  Accepted steps: grv_multi_frame_stats().accepted_steps */
int grv_render_frame_wgsl_multi_device(grv_multi *m, const GrvWgslParams *p, float *d_rgba,
                                       void *root_stream);

/* ---- post chain: device RGBA f32 images [height][width][4], row-major ----
 * Texture fetches are GL LINEAR + CLAMP_TO_EDGE with f32 weights; half_storage != 0 rounds every
 * stored channel through binary16 as the reference's RGBA16F render targets do. */
typedef struct { /* src/shaders/postprocess/reprojection.glsl.ts:44-116 uniforms */
    uint32_t width, height;
    float blend_factor;     /* u_blendFactor: 0.75 from src/rendering/webgl/renderer.ts:383 */
    int32_t camera_moving;  /* u_cameraMoving */
    int32_t half_storage;
    int32_t arith;          /* GRV_ARITH_STRICT (shader operation order) / GRV_ARITH_FAST */
} GrvTaaParams;
/* ReprojectionManager.resolve (src/rendering/reprojection.ts:196-262): out = history-blended frame */
int grv_post_taa_resolve(grv_engine *e, const GrvTaaParams *p, const float *d_current,
                         const float *d_history, float *d_out, void *stream);
/* effective blend of resolve(): clamp(0.9 - 6 v, 0.05, 0.9) when v > 0.001 (reprojection.ts:241-245) */
float grv_taa_effective_blend(float blend_factor, float camera_velocity_magnitude);

typedef struct { /* src/shaders/postprocess/ataa.wgsl.ts:29-86 (CameraUniforms, types.wgsl.ts:6-17) */
    uint32_t width, height;
    float inv_view[16], inv_proj[16], prev_view_proj[16], position[3]; /* column-major */
    int32_t half_storage;
    int32_t arith;
} GrvAtaaParams;
int grv_post_ataa_resolve(grv_engine *e, const GrvAtaaParams *p, const float *d_current,
                          const float *d_history, float *d_out, void *stream);

/* Verification hook (no device): the 24 coefficients the FAST ATAA resolve evaluates its reprojected
 * history tap from -- the shader's chain ndc -> inv_proj -> divide -> normalise -> inv_view ->
 * position + 12 dir -> prev_view_proj -> divide -> uv -> texel (ataa.wgsl.ts:60-75) folded in f64 into
 * seven linear forms of the pixel coordinates: out[3 q + {0, 1, 2}] = (per px, per py, constant) of
 * q = vt.x, vt.y, vt.z, vt.w, X, Y, W, then out[21..23] = kx, ky, kw; with s = 12 sign(vt.w) / |vt.xyz|
 * the tap sits at texel ((kx + s X) / (kw + s W), (ky + s Y) / (kw + s W)). */
int grv_ataa_reproj_fold(const GrvAtaaParams *p, float out24[24]);

typedef struct { /* src/rendering/bloom.ts:23-39 BloomConfig */
    uint32_t width, height;
    float intensity;      /* 0.5 */
    float threshold;      /* 0.8 */
    int32_t blur_passes;  /* 2 */
    int32_t half_storage;
    int32_t arith;
} GrvBloomParams;
void grv_bloom_params_default(uint32_t width, uint32_t height, GrvBloomParams *p);
/* BloomManager.applyBloomToTexture (bloom.ts:443-583, renderScale 1): bright pass at w/2 x h/2,
 * blur_passes x (H, V) 9-tap Gaussian at w/4 x h/4, combine + ACES + gamma at w x h */
int grv_post_bloom(grv_engine *e, const GrvBloomParams *p, const float *d_scene, float *d_out,
                   void *stream);

/* ---- the reference's two renderers: the callers of the path ----
 * Frame-to-frame state (history ping-pong, frame counter) lives in the engine, as it lives in
 * the renderer objects upstream.  Output: device RGBA f32 [height][width][4], the image the
 * renderer presents (8-bit swap-chain quantisation is not applied). */
/* WebGPURenderer.render (src/rendering/webgpu/renderer.ts:280-411): compute march with the
 * Halton jitter of its own frame counter (compute.wgsl.ts:134-157) -> rgba16float compute
 * texture -> ATAA resolve against the history ping-pong -> Reinhard blit.
 * camera_uniforms: the 352-byte CameraUniforms block, physics_params: the 32-byte PhysicsParams
 * block exactly as writeCameraUniforms / writePhysicsParams fill them (src/types/webgpu.ts:67-116;
 * frame_index in the block is overridden by the renderer's counter, renderer.ts:303).
 * max_steps: the pipeline's MAX_STEPS override constant (renderer.ts:188, default 150). */
int grv_webgpu_render(grv_engine *e, const float camera_uniforms[88], const float physics_params[8],
                      int32_t max_steps, int32_t arith, float *d_screen, void *stream);
/* WebGLRenderer.render (src/rendering/webgl/renderer.ts:173-422): fragment shader with
 * ENABLE_LINEAR_OUTPUT into the RGBA16F scene target -> ReprojectionManager.resolve(scene, 0.75,
 * cameraMoving) -> BloomManager.applyBloomToTexture (features.bloom) or drawTextureToScreen; both
 * end in ACES + gamma (bloom.glsl.ts:117-123).  p->tone_map is ignored (the post chain owns it). */
int grv_webgl_render(grv_engine *e, const GrvGlslParams *p, int32_t bloom_enabled,
                     int32_t camera_moving, float *d_screen, void *stream);
/* same, presenting into host memory (one PCIe copy of the final image; N-API / ctypes hosts) */
int grv_webgpu_render_host(grv_engine *e, const float camera_uniforms[88], const float physics_params[8],
                           int32_t max_steps, int32_t arith, float *screen);
int grv_webgl_render_host(grv_engine *e, const GrvGlslParams *p, int32_t bloom_enabled,
                          int32_t camera_moving, float *screen);
/* drop the history textures and frame counters (renderer resize / re-creation) */
void grv_renderer_reset(grv_engine *e);
uint32_t grv_renderer_frame_count(const grv_engine *e);

/* ---- device images: what a frame loop above the ABI holds between passes (ABI 8) ----
 * The reference's per-frame callers never hold host pixels: WebGPURenderer.render keeps the compute
 * pass's output in `computeTexture`, resolves it against the history textures and blits
 * (src/rendering/webgpu/renderer.ts:280-411); WebGLRenderer.render goes scene target -> reprojection ->
 * bloom -> screen (src/rendering/webgl/renderer.ts:173-422); the worker moves the 8 KB SAB block only
 * (src/workers/physics.worker.ts:111-176).  A grv_image is that texture: W x H RGBA f32 in HBM on the
 * engine's device, with a stream of its own.  Frames and post passes INTO an image are queued on the
 * image's stream and the call returns at once; a host that alternates two images keeps two frames in
 * flight.  Pixels cross PCIe only in grv_image_read*.  An image survives the engine that created it
 * (it holds no pointer to it); images are consumed by engines on the same device only.
 *
 * Thread rule: engine calls are single-threaded as everywhere in this ABI; grv_image_read_async /
 * _wait / _query / _read / _frame_stats touch the image alone, so one thread may wait for (or read) an
 * image while another queues the next frame into ANOTHER image through the engine. */
typedef struct grv_image grv_image;
/* a new image is black (zero-filled on its own stream, waited for: the call returns with the stream's hardware queue
 * set up, so the first frame into it does not pay for that in the middle of a burst) */
int grv_image_create(grv_engine *e, uint32_t width, uint32_t height, grv_image **out);
/* an image whose producers are queued on `stream_of`'s compute stream instead of one of its own:
 * images sharing a stream are written in queue order (frame i+1's kernels start when frame i's have
 * finished), while every image still reads back on a copy stream of its own -- the rotation a host uses
 * to move every frame to host memory: frame i's D2H runs under frame i+1's kernels.  (Images with
 * streams of their own are written concurrently: two frames share the chip, the drain of one launch
 * under the head of the next -- what the short 1080p marches want.) */
int grv_image_create_shared(grv_engine *e, uint32_t width, uint32_t height, grv_image *stream_of,
                            grv_image **out);
void grv_image_destroy(grv_image *img); /* waits for the work queued on it */
uint32_t grv_image_width(const grv_image *img);
uint32_t grv_image_height(const grv_image *img);
size_t grv_image_bytes(const grv_image *img);
float *grv_image_data(grv_image *img);   /* device pointer, [height][width][4] f32 */
void *grv_image_stream(grv_image *img);  /* hipStream_t its producers are queued on */
const char *grv_image_last_error(const grv_image *img);
/* grv_render_frame_device / grv_render_frame_glsl / grv_render_frame_wgsl with the image as the RGBA
 * target (whole frames: tile_world 0 or 1; p->width x p->height must be the image's size) */
int grv_render_frame_image(grv_engine *e, const GrvCamera *cam, const GrvRenderParams *p, grv_image *img);
int grv_render_frame_glsl_image(grv_engine *e, const GrvGlslParams *p, grv_image *img);
int grv_render_frame_wgsl_image(grv_engine *e, const GrvWgslParams *p, grv_image *img);
/* the two renderers presenting into an image; successive frames are ordered through the engine's
 * history targets whatever streams their images own */
int grv_webgl_render_image(grv_engine *e, const GrvGlslParams *p, int32_t bloom_enabled,
                           int32_t camera_moving, grv_image *screen);
int grv_webgpu_render_image(grv_engine *e, const float camera_uniforms[88], const float physics_params[8],
                            int32_t max_steps, int32_t arith, grv_image *screen);
/* post passes on images: queued on `out`'s stream behind the producers of the inputs; a later
 * producer of an input waits for this pass */
int grv_post_bloom_image(grv_engine *e, const GrvBloomParams *p, grv_image *scene, grv_image *out);
int grv_post_taa_resolve_image(grv_engine *e, const GrvTaaParams *p, grv_image *current,
                               grv_image *history, grv_image *out);
/* D2H of the first `elems` floats behind the image's last producer, on the image's own copy stream
 * (page-locked `host` memory: one DMA at the PCIe rate).
 *   grv_image_read        blocking: waits for the producers on the host, queues the copy, waits for it.  The
 *                         form a host runs on a worker thread (image-only call) while its main thread queues
 *                         the next frames into OTHER images; no producer may be queued into THIS image until
 *                         it has returned.
 *   grv_image_read_async  queues the copy behind the producers (an event wait on the copy stream) and
 *                         returns; a later producer of the image waits for it on the device.  With more live
 *                         streams than the runtime has hardware queues (4 by default) that event wait can
 *                         sit in front of another stream's kernels: prefer grv_image_read on a worker thread.
 *   grv_image_wait        blocks until the image's last producer and last read have finished (not for later
 *                         work of images sharing its stream); grv_image_query: 1 finished, 0 busy, < 0 = -status */
int grv_image_read_async(grv_image *img, float *host, size_t elems);
int grv_image_wait(grv_image *img);
int grv_image_query(grv_image *img);
int grv_image_read(grv_image *img, float *host, size_t elems);
/* counters of the frame that last wrote the image (copied behind its last kernel; with
 * grv_stats_accumulate: the running sums at that point).  Waits for the image only.  Event times and
 * `launches` are not part of it (0). */
int grv_image_frame_stats(grv_image *img, GrvFrameStats *stats);

/* camera helpers (gl-matrix lookAt/perspective as src/components/canvas/WebGPUCanvas.tsx:143-157) */
void grv_camera_look_at(const double eye[3], const double target[3], const double up[3],
                        double fovy_rad, double aspect, GrvCamera *cam);
/* from the 352-byte f32 CameraUniforms block (src/types/webgpu.ts:25) */
void grv_camera_from_uniforms(const float *camera_uniforms_88f, GrvCamera *cam);
void grv_render_params_default(uint32_t width, uint32_t height, GrvRenderParams *p);
void grv_options_default(GrvOptions *o); /* integrator.rs:35-47 */

/* ---- LUTs: generate_spectrum_lut lib.rs:128-136 (physics/spectrum.rs:76-102) ---- */
int grv_generate_spectrum_lut(grv_engine *e, size_t width, size_t height, double max_temp,
                              float *out_host);
int grv_generate_spectrum_lut_device(grv_engine *e, size_t width, size_t height,
                                     double max_temp, float *d_out, void *stream);

/* ---- the STRICT contract's transcendental functions, evaluated on the device ----
 * Rust's f64::sin/cos/powf (kerr.rs:415, integrator.rs:93,97 ...) take their last bit from the
 * linked libm; the STRICT kernels use written-out fdlibm-lineage routines instead (csrc/
 * strict_libm.hpp) so that a STRICT result is a pure function of its inputs.  This entry point
 * exposes them for verification: out[i] = op(x[i] [, y[i]]), host buffers. */
enum { GRV_MATH_SINCOS_SIN = 0, GRV_MATH_SINCOS_COS = 1, GRV_MATH_SIN = 2, GRV_MATH_COS = 3,
       GRV_MATH_POW = 4, GRV_MATH_EXP = 5, GRV_MATH_ATAN = 6, GRV_MATH_LOG = 7, GRV_MATH_ACOS = 8,
       GRV_MATH_ATAN2 = 9 /* atan2(x[i], y[i]) */,
       GRV_MATH_DIV = 10 /* x[i] / y[i] as the compiler expands it on the device (correctly rounded) */,
       GRV_MATH_DIV_SHARED = 11 /* x[i] / y[i] through the STRICT Kerr-Schild kernels' shared-reciprocal form
                                   (csrc/kerr_device.hpp SharedDiv): the same bits as GRV_MATH_DIV whenever
                                   both operands are zero or moderate in magnitude, which is the only case in
                                   which the kernels use it */,
       GRV_MATH_DIV_NOFIX = 12 /* the same without the closing v_div_fixup (SharedDivNoFixup): the IEEE quotient
                                  for y[i] > 0 and x[i] = +0 or non-zero, both finite and moderate */,
       GRV_MATH_RCP_R2 = 13 /* the refined reciprocal SharedDiv::prep(x[i]) holds (rcp seed + two Newton steps):
                               lets a test assert that the literal reciprocals of ConstDen are the device's */,
       GRV_MATH_DIV_CONST = 14 /* x[i] / y[i] through ConstDen for y[i] in {2197, 216, 513, 4104, 27, 2565, 40}
                                  (the Fehlberg tableau's denominators); any other y[i] yields NaN */,
       GRV_MATH_F32 = 16 /* or-ed in: the f32 form (float)op((double)(float)x) of the shader-order kernels */ };
int grv_strict_math(grv_engine *e, int op, size_t n, const double *x, const double *y, double *out);
/* The STRICT Kerr-Schild right-hand side (get_state_derivative, geodesic/hamiltonian.rs:13-35 over
 * kerr.rs:412-499) of n states [t, r, theta, phi, p_t, p_r, p_theta, p_phi] through one of the three
 * division forms the STRICT kernels choose between per wave (csrc/kerr_device.hpp): verification hook,
 * host buffers.  out[7 i ..] = dt, dr, dtheta, dphi, dp_r, dp_theta, and the form that ran -- the one
 * asked for where its operand guard admits the state, else the next more general one.  Every form must
 * return the same bits. */
enum { GRV_RHS_FORM_IEEE = 0 /* the compiler's `/` */, GRV_RHS_FORM_SHARED = 1 /* SharedDiv */,
       GRV_RHS_FORM_NOFIXUP = 2 /* SharedDivNoFixup */ };
int grv_strict_rhs_probe(grv_engine *e, int form, size_t n, const double *states, double *out);
/* the same routines compiled for the host (they also serve the engine's host-side closed forms):
 * no device needed.  op as above; GRV_MATH_SINCOS_* evaluate sin / cos. */
int grv_strict_math_host(int op, size_t n, const double *x, const double *y, double *out);

/* ---- disk / shadow helpers next to the path ---- */
/* generate_disk_lut lib.rs:107-110 (physics/disk.rs:175-201): 512 normalised temperatures */
int grv_generate_disk_lut(grv_engine *e, float *out512);
/* compute_disk_flux lib.rs:198-200 (physics/disk.rs:90-151, m_dot = 1) */
double grv_compute_disk_flux(const grv_engine *e, double r);
/* compute_shadow_curve lib.rs:161-169 (physics/shadow.rs:81-183): (alpha, beta) pairs as f32.
 * The curve has n_points points off axis and 2 * n_points for an on-axis observer
 * (|sin theta_obs| < 1e-10, shadow.rs:96-113).  Returns the number of POINTS of the whole curve;
 * at most out_capacity floats are written (whole pairs only), so out == NULL / out_capacity == 0
 * is a size query and a too-small buffer is never overrun. */
size_t grv_compute_shadow_curve(const grv_engine *e, double theta_obs, size_t n_points, float *out,
                                size_t out_capacity);
/* compute_shadow_radius lib.rs:172-174 ; compute_shadow_shift lib.rs:178-195 */
double grv_compute_shadow_radius(const grv_engine *e);
int grv_compute_shadow_shift(const grv_engine *e, double theta_obs, float out2[2]);

/* get_disk_lut_ptr lib.rs:112-114: the engine-owned copy of the last generate_disk_lut
 * result (512 floats, zeros before the first call) */
const float *grv_get_disk_lut_ptr(const grv_engine *e);

/* ---- spacetime read-outs (viz helpers beside the path): lib.rs:139-159, 214-306 over
 * gravitas-core/src/spacetime/{curvature,lightcone,frame_drag,embedding}.rs ---- */
double grv_compute_kretschner(const grv_engine *e, double r, double theta);       /* lib.rs:214 */
double grv_compute_light_cone_tilt(const grv_engine *e, double r, double theta);  /* lib.rs:239 */
double grv_compute_frame_drag_omega(const grv_engine *e, double r, double theta); /* lib.rs:268 */
double grv_compute_flamm_height(const grv_engine *e, double r);                   /* lib.rs:297 */
double grv_compute_proper_distance(const grv_engine *e, double r1, double r2, size_t n_steps); /* :303 */
#define GRV_FIELD_CURVATURE 0  /* generate_curvature_field  lib.rs:220-236 */
#define GRV_FIELD_TILT 1       /* generate_tilt_field       lib.rs:245-265 */
#define GRV_FIELD_FRAME_DRAG 2 /* generate_frame_drag_field lib.rs:274-294 */
/* (r, theta, value) f32 triples, radial index slowest; `out` holds 3*n_radial*n_polar floats */
int grv_generate_field(grv_engine *e, int field, double r_min, double r_max, size_t n_radial,
                       size_t n_polar, float *out);
/* generate_embedding_mesh lib.rs:139-150: xyz f32 vertices, 3*n_radial*n_angular floats */
int grv_generate_embedding_mesh(grv_engine *e, double r_min, double r_max, size_t n_radial,
                                size_t n_angular, float *out);
/* generate_ergosphere_mesh lib.rs:153-157: xyz f32 vertices, 3*n_polar*n_azimuthal floats */
int grv_generate_ergosphere_mesh(grv_engine *e, size_t n_polar, size_t n_azimuthal, float *out);

/* ---- SAB protocol: lib.rs:36-40, 74, 116-126, 308-419 (offsets in f32 elements) ---- */
const float *grv_get_sab_ptr(const grv_engine *e);
void grv_get_sab_layout(size_t out5[5]);
int grv_attach_sab(grv_engine *e, float *ptr);                     /* attach_sab lib.rs:74 */
void grv_set_camera_state(grv_engine *e, double px, double py, double pz); /* lib.rs:120 */
void grv_set_auto_spin(grv_engine *e, int enabled);                /* lib.rs:124 */
int grv_tick_sab(grv_engine *e, double dt_override);               /* lib.rs:308-409 */

#ifdef __cplusplus
}
#endif
#endif
