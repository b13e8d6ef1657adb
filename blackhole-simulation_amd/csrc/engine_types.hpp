// engine_types.hpp -- plain structs shared by the host engine and the device
// translation units (kernels_strict.hip / kernels_fast.hip), plus the launcher
// entry points those units export.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/gravitas_abi.h"

namespace grvhip {

constexpr int kBlock = 256;
// threads per block of the f32 march kernels (WGSL compute march in its three forms, GLSL fragment
// shader): ONE wave.  A march wave's lanes run 35..1024 steps and its slot (registers, wave slot) goes
// back to the dispatcher only when its whole block has ended; with four-wave blocks the packed march
// measured 287 G ray-steps/s on the 8K config-4 frame, with one-wave blocks 308 G (A/B on one box,
// profiles/r03_ab_march_block.jsonl).  The f64 segment kernel chooses per launch (geodesic_kernels.hpp:
// segment_block_threads): one-wave blocks for launches with a dispatch order (whole frames of 65 536 rays and more) and
// for one-launch launches of 1.5 M rays and more (+1.1 %), kBlock otherwise.
#ifndef GRV_MARCH_BLOCK
#define GRV_MARCH_BLOCK 64
#endif
constexpr int kMarchBlock = GRV_MARCH_BLOCK;
// Dispatch order of a frame's blocks.  Workgroups start in blockIdx order, and a frame's long rays
// (photon ring, disk) sit in the middle rows of the image, i.e. in the middle of a rank's block list:
// centre-out order starts them first, so the last blocks to start hold the short rays of the image's
// edge rows and the launch's tail is as short as it can be.  Storage is untouched: a block only
// changes WHICH slots it works on, not where they live.  Measured A/B on one box
// (profiles/r03_ab_centre_out.jsonl): the packed f32 march of config 4, whose 0.013 % of rays at the
// 1024-step budget run seven times as long as the rest, gains 2.1 % on the whole 8K frame (311.0 ->
// 317.4 G ray-steps/s) and 1 % on an eighth of it with two frames in flight; the f64 RKF45 frame
// (longest wave 3x the median) loses 0.5 % at N = 1 and gains 1.4 % on an eighth -- so the f32 marches
// take it, and the f64 segment kernel takes it for a rank's share of a split frame only
// (SegmentParams::block_order, set by grv_render_frame_device when tile_world > 1; measured on the
// shares of the strong 4K split with two frames in flight: N = 2 14.10 -> 13.84 ms, N = 4 7.07 -> 7.02,
// N = 8 3.64 -> 3.59) and keeps the natural order for a whole frame -- until the measured dispatch order (round 6,
// SegmentParams::order) replaced both for every f64 frame of 65 536 rays and more, rank shares included (N = 8: 3.49 ->
// 3.43 ms with two frames in flight, 3.87 -> 3.49 ms with one; centre-out is what a share runs under
// GRV_SCHEDULE_SLOT_ORDER).  The GLSL
// fragment march was measured too and loses 2-12 % (its long rays are the disk-slab samplers, which
// contend when they all start together): natural order there.  A prime stride through the block list
// (consecutive starts a quarter of the image apart) was measured as well and loses everywhere: f64
// -1.5 %, packed march -2 % against centre-out, GLSL up to -20 % (neighbouring tiles share the noise /
// star texels' cache lines) -- lib_f2 / lib_p2 / lib_g2 in the same file.  An XCD-contiguous deal (each
// XCD takes a contiguous eighth of the block list) loses 30 % on the GLSL march: the eighths differ in
// work and the XCDs finish at different times.
#ifndef GRV_CENTRE_OUT
#define GRV_CENTRE_OUT 1
#endif
#ifndef GRV_CENTRE_OUT_F64
#define GRV_CENTRE_OUT_F64 0
#endif
__device__ __forceinline__ uint32_t centre_out_block(uint32_t b, uint32_t nb) {
    const uint32_t half = nb >> 1;
    return (b & 1u) ? (half - 1u - (b >> 1)) : (half + (b >> 1)); // bijection of [0, nb)
}
__device__ __forceinline__ uint32_t dispatch_block(uint32_t b, uint32_t nb) {
#if GRV_CENTRE_OUT
    return centre_out_block(b, nb);
#else
    return b;
#endif
}
__device__ __forceinline__ uint32_t dispatch_block_f64(uint32_t b, uint32_t nb, uint32_t order) {
#if GRV_CENTRE_OUT_F64
    return centre_out_block(b, nb);
#else
    return order ? centre_out_block(b, nb) : b;
#endif
}
// ---- measured-cost dispatch order of the dispatched FAST marches ------------------------------------------
// Workgroups start in blockIdx order.  `order` (a permutation of the frame's blocks, device memory) says
// which block a workgroup takes; `cost` receives every block's duration on the constant-rate clock.  After
// the march a one-workgroup counting sort turns the costs into the next frame's order, longest first
// (longest-processing-time-first list scheduling: the launch's tail is then made of the cheapest blocks).
// A renderer's frames resemble their predecessors, so last frame's durations are this frame's forecast; any
// permutation gives the same image -- only the launch's length changes.  Either pointer may be null
// (natural / centre-out order, nothing recorded).
struct MarchSched {
    const uint32_t *order;
    uint32_t *cost;
};
hipError_t launch_march_order_identity(uint32_t *order, uint32_t *cost, uint32_t n, hipStream_t s);
hipError_t launch_march_rank(const uint32_t *cost, uint32_t *order, uint32_t n, hipStream_t s);
// slot lists of the first n_head waves of `order` (64 slots each) and of the remaining ones (kernels_strict.hip)
hipError_t launch_split_order(const uint32_t *order, uint32_t n_waves, uint32_t n_head, uint32_t *head, uint32_t *rest,
                              hipStream_t s);
constexpr int kMaxCrossRec = 4;

// flags word: bits 0-2 termination, bit 3 forced-min-step pending, bits 4-7 crossing
// count, bit 8 slot holds a real ray.
constexpr uint32_t kFlagTermMask = 0x7u;
constexpr uint32_t kFlagForced = 0x8u;
constexpr uint32_t kFlagCrossShift = 4;
constexpr uint32_t kFlagCrossMask = 0xF0u;
constexpr uint32_t kFlagValid = 0x100u;

struct RayWorkspace {
    double *t, *r, *th, *ph, *pr, *pth; // evolving components
    double *pt, *pph;                   // constants of motion (E = -p_t, L_z = p_phi)
    double *h;                          // carried adaptive step
    double *drift;                      // max |H|
    double *rc;                         // [kMaxCrossRec][n] disk-plane crossing radii
    uint32_t *steps, *tries, *flags;
    uint32_t n; // slots
};

struct SegmentParams {
    double M, a, a2;
    double horizon_limit; // r_+ * 1.001
    double escape_radius;
    double tolerance;
    double inv_tolerance; // FAST contract: err * (1/tol)
    double step_size;     // RK4 / symplectic
    uint32_t max_steps;
    uint32_t renorm_interval;
    uint32_t max_tries; // per launch
    uint32_t final_launch; // segment kernel: no launch follows -- a ray still live after max_tries
                           // (the hard bound of engine.hip try_bound) is ended as TERM_MAXSTEPS
    uint32_t try_cap;      // refill / single-ray kernels: the same bound on a ray's total tries
    // disk-plane crossing recorder (shading)
    int32_t shading;
    double disk_inner, disk_outer;
    uint32_t max_crossings;
    uint32_t block_order; // segment kernel without a dispatch order: 0 = blocks in slot order, 1 = centre-out (a rank share under GRV_SCHEDULE_SLOT_ORDER)
    // segment kernel, one-launch schedule of whole frames: the block a workgroup takes (a permutation of the
    // launch's one-wave blocks, longest first by LAST frame's per-wave tries -- finalize_frame_kernel writes the
    // costs, march_rank_kernel sorts them); null = block_order decides.  Any permutation gives the same frame.
    const uint32_t *order;
};
// one-launch launches WITHOUT a dispatch order (the compacting schedule's head start; frames under
// GRV_SCHEDULE_SLOT_ORDER) start one-wave blocks from this many rays on (geodesic_kernels.hpp segment_block_threads)
constexpr uint32_t kSegOneWaveMinRays = 3u << 19; // 1 572 864
// Frames (whole ones and a rank's share of a split one) take a measured dispatch order -- and with it one-wave blocks: the
// order's entries are one-wave blocks -- from this many rays on.  Until the resolution sweep of round 6 this was kSegOneWaveMinRays too, and every frame below
// 1080p ran four-wave blocks in slot order: 720p 44.1 -> 50.5 G ray-steps/s, 540p 40.1 -> 47.6, 360p 31.7 -> 37.9, 720p at
// r0 = 10 M 34.8 -> 43.6 (profiles/r06_ab_small_frame_order.jsonl).  Below 65 536 rays (1 024 waves, a third of the
// chip's wave slots) there is nothing to order.
#ifndef GRV_SEG_ORDER_MIN_RAYS
#define GRV_SEG_ORDER_MIN_RAYS 65536u
#endif
constexpr uint32_t kSegOrderMinRays = GRV_SEG_ORDER_MIN_RAYS;
// cost entry of a wave whose slowest ray took `tries` integrator tries (march_rank_kernel buckets by cost >> 7)
constexpr uint32_t kWaveCostShift = 6;

struct FrameGeom {
    uint32_t width, height;
    uint32_t tiles_x, tiles_y; // tiles_x = row pitch of the tile ids (engine.hip: tile_pitch)
    uint32_t tile_world, tile_rank;
    uint32_t n_tiles_local;
};

struct CameraDev {
    double pos[3];
    double inv_view[16];
    double inv_proj[16];
    double off[2];
    // compute.wgsl.ts:172-176, per-frame constants of the pixel -> state map (host-evaluated)
    double r0, theta0, phi0, st, ct, sp, cp;
};

constexpr uint32_t kDiskLutWidth = 512; // lut_width of generate_disk_lut (lib.rs:65)

struct ShadeParams {
    double M, spin;
    double disk_inner, disk_temp, disk_opacity, exposure;
    uint32_t lut_w, lut_h;
    double lut_max_temp;
    uint32_t lds_row0, lds_rows; // LUT rows staged in LDS
    // radial temperature profile (GRV_DISK_PROFILE_*): the Page-Thorne table spans
    // [pt_rin, pt_rout] = [prograde ISCO, 50 M] (physics/disk.rs:176-177)
    uint32_t disk_profile;
    double pt_rin, pt_rout;
};

// uniforms of the two f32 shader kernels (mirrors GrvWgslParams / GrvGlslParams)
struct WgslParams {
    float inv_view[16];
    float inv_proj[16];
    float position[3];
    float mass, spin;
    float jitter[2];
    int32_t max_steps;
    int32_t stars; // escape-branch star hash on/off
};
struct GlslParams {
    float mass, spin, zoom;
    float mouse[2];
    float disk_size, disk_scale_height, disk_density, disk_temp;
    float lensing_strength, time, turbulence;
    int32_t max_ray_steps, tone_map;
    uint32_t features; // GRV_GLSL_* bits
    int32_t quality;   // 0: RAY_QUALITY_LOW/OFF indicator path
    float show_redshift, show_kerr_shadow, debug;
    float cam_pos[3], cam_quat[4];
    float shadow_count;
    float shadow_curve[64][2];
    const uint8_t *noise_r; // 256x256 R channel of u_noiseTex (device)
    const uint8_t *blue_r;  // 256x256 R channel of u_blueNoiseTex (device)
    const float *noise_f;   // the same noise plane as f32 texel values (byte / 255.0f), FAST lattice noise
};

// single-ray entry (grv_integrate_ray_relativistic): arguments by value, result in pinned host memory
struct SingleRayIn {
    double v[8]; // GeodesicState: t, r, theta, phi, p_t, p_r, p_theta, p_phi (geodesic/mod.rs:23-30)
};
struct SingleRayOut {
    double state[8];
    double drift;
    uint32_t steps, tries, term;
    uint32_t seq; // written last (system scope): the call's sequence number
    // the try loop on the device's own clocks (s_memtime: shader cycles; s_memrealtime: the constant
    // 100 MHz counter): what one accepted step costs without the launch and the PCIe round trip
    unsigned long long loop_cycles, loop_ticks;
};

// The f32 marches add their step total once per one-wave block.  32 400 blocks of a 1080p frame adding to ONE address
// retire at ~13 ns apiece: 0.43 ms when they all come at once -- which is the whole frame of a close-up view (25-50
// steps per pixel; profiles/EXPERIMENTS.md J, X).  They add to one of kStepParts slots by block index instead (one
// 128-byte line each: atomics on one line still queue behind each other); readers sum (stats_total_steps).
constexpr uint32_t kStepParts = 32, kStepPartStride = 16; // 32 slots, 16 u64 = 128 B apart
struct FrameStatsDev {
    unsigned long long accepted_steps, rkf_tries, term_count[5], crossings, rays;
    unsigned long long max_drift_bits;
    unsigned long long pad_[6];                               // steps_part starts on a 128-byte line of the block
    unsigned long long steps_part[kStepParts * kStepPartStride];
};
static_assert(offsetof(FrameStatsDev, steps_part) % 128 == 0, "FrameStatsDev: the step slots sit on lines of their own");
inline unsigned long long stats_total_steps(const FrameStatsDev &d) {
    unsigned long long t = d.accepted_steps;
    for (uint32_t k = 0; k < kStepParts; ++k) t += d.steps_part[k * kStepPartStride];
    return t;
}


// ---- launchers (kernels_strict.hip: -ffp-contract=off) ----
hipError_t launch_segment_strict(int kind, int method, const RayWorkspace &ws,
                                 const SegmentParams &P, const uint32_t *live_in, uint32_t n_live,
                                 uint32_t *live_out, uint32_t *live_out_count, hipStream_t s);
hipError_t launch_refill_strict(int kind, int method, const RayWorkspace &ws, const SegmentParams &P,
                                uint32_t *cursor, int n_cu, hipStream_t s, uint32_t block_threads = 0);
hipError_t launch_path_strict(int kind, int method, const RayWorkspace &ws, const SegmentParams &P,
                              const double *states_in, double *paths, uint32_t *counts, uint32_t max_points,
                              hipStream_t s);
hipError_t launch_single_ray(int kind, const SegmentParams &P, const SingleRayIn &in, double h0,
                             SingleRayOut *out_pinned, uint32_t seq, hipStream_t s);
hipError_t launch_single_ray_fast(int kind, const SegmentParams &P, const SingleRayIn &in, double h0,
                                  SingleRayOut *out_pinned, uint32_t seq, hipStream_t s); // kernels_fast.hip
hipError_t launch_init_states(int kind, const RayWorkspace &ws, const SegmentParams &P,
                              const double *states, double h0, int adaptive, hipStream_t s);
hipError_t launch_init_pixels(int kind, const RayWorkspace &ws, const SegmentParams &P,
                              const FrameGeom &G, const CameraDev &cam, double h0, int adaptive,
                              hipStream_t s);
// block_threads (0 = the default four-wave blocks): worker-sized batches on the control stream start ONE-wave blocks --
// beside a frame kernel at full occupancy a block is dispatched when its waves fit, and a single wave fits as soon as
// one frame wave leaves a SIMD, a four-wave block only when one leaves on all four SIMDs of a CU at once (i.e. at the
// frame's drain: such a batch waited for two of three queued 4K frames)
hipError_t launch_finalize_batch(const RayWorkspace &ws, double *out_states, uint32_t *out_steps,
                                 uint8_t *out_term, double *out_drift, FrameStatsDev *st,
                                 hipStream_t s, uint32_t block_threads = 0);
hipError_t launch_finalize_frame(const RayWorkspace &ws, const FrameGeom &G, const ShadeParams &S,
                                 int shading, const float *lut, const float *disk_lut, float *out_rgba,
                                 double *out_states, uint32_t *out_steps, uint8_t *out_term,
                                 double *out_drift, FrameStatsDev *st, int n_blocks,
                                 hipStream_t s, uint32_t *wave_cost = nullptr);
hipError_t launch_pack_half(const float *rgba, void *half4, size_t n_px, hipStream_t s);
hipError_t launch_widen_half(const void *half4, float *rgba, size_t n_px, hipStream_t s);
hipError_t launch_unpack_tiles_half(const FrameGeom &G, const void *packed_half4, float *image, hipStream_t s);
hipError_t launch_unpack_tiles(const FrameGeom &G, const void *packed, void *image,
                               uint32_t words_per_pixel, hipStream_t s);
hipError_t launch_wgsl_symplectic(const FrameGeom &G, const WgslParams &P, float *out_rgba,
                                  uint32_t *out_steps, unsigned long long *total_steps,
                                  uint32_t n_slots, hipStream_t s);
hipError_t launch_glsl_fragment(const FrameGeom &G, const GlslParams &P, float *out_rgba,
                              uint32_t *out_steps, unsigned long long *total_steps,
                              uint32_t n_slots, hipStream_t s);
hipError_t launch_strict_rhs_probe(int form, uint32_t n, double M, double a, const double *states, double *out,
                                   hipStream_t s);
hipError_t launch_strict_math(int op, uint32_t n, const double *x, const double *y, double *out,
                              hipStream_t s);
hipError_t launch_spectrum_lut(float *out, uint32_t width, uint32_t height, double max_temp,
                               hipStream_t s);
// ---- post chain (post_kernels.hpp, in kernels_strict.hip) ----
struct AtaaCameraHost {
    float inv_view[16], inv_proj[16], prev_view_proj[16], position[3];
};
hipError_t launch_taa_resolve(uint32_t w, uint32_t h, const float *current, const float *history,
                              float blend_factor, int camera_moving, int half_storage, float *out,
                              hipStream_t s);
// The FAST ATAA resolve's reprojection chain folded once per launch (post_fast_kernels.hpp, AtaaReproj):
// c[q] = (per px, per py, constant) of vt.x, vt.y, vt.z, vt.w and the texel-x / texel-y / w combinations
// of the clip rows (the part multiplied by s = 12 sign(vt.w) / |vt.xyz|), k = their constant parts.
// Plain host arithmetic in f64 (matrices column-major: m[col * 4 + row]); grv_ataa_reproj_fold exposes
// it to the CPU tests.
struct AtaaReprojHost {
    float c[7][3];
    float k[3];
};
inline void ataa_reproj_fold(const AtaaCameraHost &cam, uint32_t w, uint32_t h, AtaaReprojHost &rp) {
    auto at = [](const float *m, int r, int col) { return (double)m[col * 4 + r]; };
    double A[4][3], B[4][4], K[4];
    for (int r = 0; r < 4; ++r) {
        for (int j = 0; j < 3; ++j) {
            A[r][j] = 0.0;
            for (int t = 0; t < 3; ++t) A[r][j] += at(cam.prev_view_proj, r, t) * at(cam.inv_view, t, j);
        }
        for (int j = 0; j < 4; ++j) {
            B[r][j] = 0.0;
            for (int t = 0; t < 3; ++t) B[r][j] += A[r][t] * at(cam.inv_proj, t, j);
        }
        K[r] = at(cam.prev_view_proj, r, 3);
        for (int t = 0; t < 3; ++t) K[r] += at(cam.prev_view_proj, r, t) * (double)cam.position[t];
    }
    // rows: vt (inv_proj rows 0..3), then the texel-x, texel-y and w combinations of the clip rows:
    // x = (0.5 w clip.x + (0.5 w - 0.5) clip.w) / clip.w,  y = (-0.5 h clip.y + (0.5 h - 0.5) clip.w) / clip.w
    const double hw = 0.5 * (double)w, hh = 0.5 * (double)h;
    double M[7][4], Kc[3];
    for (int j = 0; j < 4; ++j) {
        for (int r = 0; r < 4; ++r) M[r][j] = at(cam.inv_proj, r, j);
        M[4][j] = hw * B[0][j] + (hw - 0.5) * B[3][j];
        M[5][j] = -hh * B[1][j] + (hh - 0.5) * B[3][j];
        M[6][j] = B[3][j];
    }
    Kc[0] = hw * K[0] + (hw - 0.5) * K[3];
    Kc[1] = -hh * K[1] + (hh - 0.5) * K[3];
    Kc[2] = K[3];
    // M (ndc.x, -ndc.y, 1, 1) with ndc.x = (2 px + 1) / w - 1, ndc.y = (2 py + 1) / h - 1
    for (int q = 0; q < 7; ++q) {
        rp.c[q][0] = (float)(M[q][0] * 2.0 / (double)w);
        rp.c[q][1] = (float)(-M[q][1] * 2.0 / (double)h);
        rp.c[q][2] = (float)(M[q][0] * (1.0 / (double)w - 1.0) - M[q][1] * (1.0 / (double)h - 1.0) + M[q][2] + M[q][3]);
    }
    for (int q = 0; q < 3; ++q) rp.k[q] = (float)Kc[q];
}
hipError_t launch_ataa_resolve(uint32_t w, uint32_t h, const AtaaCameraHost &cam, const float *current,
                               const float *history, int half_storage, float *out, hipStream_t s);
// scratch: (w/2*h/2 + 2*(w/4*h/4)) float4
hipError_t launch_bloom(uint32_t w, uint32_t h, const float *scene, float threshold, float intensity,
                        int blur_passes, int half_storage, float *scratch, float *out, hipStream_t s);
size_t bloom_scratch_floats(uint32_t w, uint32_t h);
hipError_t launch_post_quantize(float *img, uint32_t n_px, hipStream_t s);
// the same launchers in the FAST contract (kernels_fast.hip)
hipError_t launch_taa_resolve_fast(uint32_t w, uint32_t h, const float *current, const float *history,
                                   float blend_factor, int camera_moving, int half_storage, float *out,
                                   hipStream_t s);
hipError_t launch_ataa_resolve_fast(uint32_t w, uint32_t h, const AtaaCameraHost &cam, const float *current,
                                    const float *history, int half_storage, float *out, hipStream_t s);
hipError_t launch_bloom_fast(uint32_t w, uint32_t h, const float *scene, float threshold, float intensity,
                             int blur_passes, int half_storage, float *scratch, float *out, hipStream_t s);
hipError_t launch_blit_reinhard_fast(uint32_t w, uint32_t h, const float *src, float *dst, hipStream_t s);
hipError_t launch_blit_reinhard(uint32_t w, uint32_t h, const float *src, float *dst, hipStream_t s);
// ---- launcher (kernels_fast.hip: -ffp-contract=fast) ----
hipError_t launch_glsl_fragment_fast(const FrameGeom &G, const GlslParams &P, float *out_rgba,
                                     uint32_t *out_steps, unsigned long long *total_steps,
                                     uint32_t n_slots, MarchSched sched, hipStream_t s);
hipError_t launch_wgsl_symplectic_pk(const FrameGeom &G, const WgslParams &P, float *out_rgba,
                                     uint32_t *out_steps, unsigned long long *total_steps,
                                     uint32_t n_slots, MarchSched sched, hipStream_t s);
// blocks (= entries of MarchSched's arrays) of the dispatched forms of the two marches for n_slots slots
uint32_t march_blocks_glsl(uint32_t n_slots);
uint32_t march_blocks_pk(uint32_t n_slots, int32_t max_steps);
hipError_t launch_wgsl_symplectic_fast(const FrameGeom &G, const WgslParams &P, float *out_rgba,
                                       uint32_t *out_steps, unsigned long long *total_steps,
                                       uint32_t n_slots, hipStream_t s);
hipError_t launch_segment_fast(int kind, int method, const RayWorkspace &ws,
                               const SegmentParams &P, const uint32_t *live_in, uint32_t n_live,
                               uint32_t *live_out, uint32_t *live_out_count, hipStream_t s);
hipError_t launch_refill_fast(int kind, int method, const RayWorkspace &ws, const SegmentParams &P,
                              uint32_t *cursor, int n_cu, hipStream_t s, uint32_t block_threads = 0);
// launches 2.. of the compacting schedule: the live list's length comes from device memory (geodesic_kernels.hpp
// integrate_compact_kernel); `blocks` four-wave blocks stride over it
hipError_t launch_compact_fast(int kind, int method, const RayWorkspace &ws, const SegmentParams &P,
                               const uint32_t *live_in, const uint32_t *live_in_count, uint32_t *live_out,
                               uint32_t *live_out_count, uint32_t *clear_count, uint32_t *feedback, uint32_t blocks,
                               hipStream_t s);
hipError_t launch_compact_strict(int kind, int method, const RayWorkspace &ws, const SegmentParams &P,
                                 const uint32_t *live_in, const uint32_t *live_in_count, uint32_t *live_out,
                                 uint32_t *live_out_count, uint32_t *clear_count, uint32_t *feedback, uint32_t blocks,
                                 hipStream_t s);
hipError_t launch_path_fast(int kind, int method, const RayWorkspace &ws, const SegmentParams &P,
                            const double *states_in, double *paths, uint32_t *counts, uint32_t max_points,
                            hipStream_t s);

} // namespace grvhip
