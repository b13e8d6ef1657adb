// kernels_strict.hip -- STRICT arithmetic contract: compiled with
// -ffp-contract=off (reference operation order, IEEE divide/sqrt, no FMA).
// Also holds the kernels that exist once: init, live-list, finalize/shade, LUT.
#define GRV_SPECIFIED_LIBM 1 // sin / cos / pow of this unit: strict_libm.hpp
#include <atomic>
#include <cstring>

#include "shader_kernels.hpp"
#include "post_kernels.hpp"

namespace grvhip {

namespace {
template <int KIND, int METHOD>
hipError_t go(const RayWorkspace &ws, const SegmentParams &P, const uint32_t *live_in,
              uint32_t n_live, uint32_t *live_out, uint32_t *live_out_count, hipStream_t s) {
    const uint32_t threads = segment_block_threads(live_out, n_live, P.order);
    const uint32_t grid = (n_live + threads - 1) / threads;
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL((integrate_segment_kernel<KIND, GRV_ARITH_STRICT, METHOD>), dim3(grid),
                       dim3(threads), 0, s, ws, P, live_in, n_live, live_out, live_out_count);
    return hipGetLastError();
}
template <int KIND>
hipError_t by_method(int method, const RayWorkspace &ws, const SegmentParams &P,
                     const uint32_t *live_in, uint32_t n_live, uint32_t *live_out,
                     uint32_t *live_out_count, hipStream_t s) {
    switch (method) {
    case GRV_METHOD_RKF45: return go<KIND, GRV_METHOD_RKF45>(ws, P, live_in, n_live, live_out, live_out_count, s);
    case GRV_METHOD_RK4: return go<KIND, GRV_METHOD_RK4>(ws, P, live_in, n_live, live_out, live_out_count, s);
    case GRV_METHOD_SYMPLECTIC: return go<KIND, GRV_METHOD_SYMPLECTIC>(ws, P, live_in, n_live, live_out, live_out_count, s);
    default: return hipErrorInvalidValue;
    }
}
} // namespace

hipError_t launch_segment_strict(int kind, int method, const RayWorkspace &ws,
                                 const SegmentParams &P, const uint32_t *live_in, uint32_t n_live,
                                 uint32_t *live_out, uint32_t *live_out_count, hipStream_t s) {
    switch (kind) {
    case GRV_METRIC_KERR_KS: return by_method<GRV_METRIC_KERR_KS>(method, ws, P, live_in, n_live, live_out, live_out_count, s);
    case GRV_METRIC_KERR_BL: return by_method<GRV_METRIC_KERR_BL>(method, ws, P, live_in, n_live, live_out, live_out_count, s);
    case GRV_METRIC_SCHWARZSCHILD: return by_method<GRV_METRIC_SCHWARZSCHILD>(method, ws, P, live_in, n_live, live_out, live_out_count, s);
    default: return hipErrorInvalidValue;
    }
}

#define GRV_REFILL_ARITH GRV_ARITH_STRICT
#define GRV_REFILL_FN launch_refill_strict
#include "refill_launch.inc"
#undef GRV_REFILL_ARITH
#undef GRV_REFILL_FN

#define GRV_COMPACT_ARITH GRV_ARITH_STRICT
#define GRV_COMPACT_FN launch_compact_strict
#include "compact_launch.inc"
#undef GRV_COMPACT_ARITH
#undef GRV_COMPACT_FN

#define GRV_PATH_ARITH GRV_ARITH_STRICT
#define GRV_PATH_FN launch_path_strict
#include "path_launch.inc"
#undef GRV_PATH_ARITH
#undef GRV_PATH_FN

hipError_t launch_single_ray(int kind, const SegmentParams &P, const SingleRayIn &in, double h0,
                             SingleRayOut *out_pinned, uint32_t seq, hipStream_t s) {
    switch (kind) {
    case GRV_METRIC_KERR_KS:
        hipLaunchKernelGGL((single_ray_kernel<GRV_METRIC_KERR_KS, GRV_ARITH_STRICT>), dim3(1), dim3(64), 0, s, P, in, h0, out_pinned, seq);
        break;
    case GRV_METRIC_KERR_BL:
        hipLaunchKernelGGL((single_ray_kernel<GRV_METRIC_KERR_BL, GRV_ARITH_STRICT>), dim3(1), dim3(64), 0, s, P, in, h0, out_pinned, seq);
        break;
    case GRV_METRIC_SCHWARZSCHILD:
        hipLaunchKernelGGL((single_ray_kernel<GRV_METRIC_SCHWARZSCHILD, GRV_ARITH_STRICT>), dim3(1), dim3(64), 0, s, P, in, h0, out_pinned, seq);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_init_states(int kind, const RayWorkspace &ws, const SegmentParams &P,
                              const double *states, double h0, int adaptive, hipStream_t s) {
    const uint32_t grid = (ws.n + kBlock - 1) / kBlock;
    if (grid == 0) return hipSuccess;
    switch (kind) {
    case GRV_METRIC_KERR_KS:
        hipLaunchKernelGGL((init_from_states_kernel<GRV_METRIC_KERR_KS>), dim3(grid), dim3(kBlock), 0, s, ws, P, states, h0, adaptive);
        break;
    case GRV_METRIC_KERR_BL:
        hipLaunchKernelGGL((init_from_states_kernel<GRV_METRIC_KERR_BL>), dim3(grid), dim3(kBlock), 0, s, ws, P, states, h0, adaptive);
        break;
    case GRV_METRIC_SCHWARZSCHILD:
        hipLaunchKernelGGL((init_from_states_kernel<GRV_METRIC_SCHWARZSCHILD>), dim3(grid), dim3(kBlock), 0, s, ws, P, states, h0, adaptive);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_init_pixels(int kind, const RayWorkspace &ws, const SegmentParams &P,
                              const FrameGeom &G, const CameraDev &cam, double h0, int adaptive,
                              hipStream_t s) {
    const uint32_t grid = (ws.n + kBlock - 1) / kBlock;
    if (grid == 0) return hipSuccess;
    switch (kind) {
    case GRV_METRIC_KERR_KS:
        hipLaunchKernelGGL((init_from_pixels_kernel<GRV_METRIC_KERR_KS>), dim3(grid), dim3(kBlock), 0, s, ws, P, G, cam, h0, adaptive);
        break;
    case GRV_METRIC_KERR_BL:
        hipLaunchKernelGGL((init_from_pixels_kernel<GRV_METRIC_KERR_BL>), dim3(grid), dim3(kBlock), 0, s, ws, P, G, cam, h0, adaptive);
        break;
    case GRV_METRIC_SCHWARZSCHILD:
        hipLaunchKernelGGL((init_from_pixels_kernel<GRV_METRIC_SCHWARZSCHILD>), dim3(grid), dim3(kBlock), 0, s, ws, P, G, cam, h0, adaptive);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_finalize_batch(const RayWorkspace &ws, double *out_states, uint32_t *out_steps,
                                 uint8_t *out_term, double *out_drift, FrameStatsDev *st,
                                 hipStream_t s, uint32_t block_threads) {
    const uint32_t bt = block_threads ? block_threads : (uint32_t)kBlock;
    uint32_t grid = (ws.n + bt - 1) / bt;
    if (grid == 0) return hipSuccess;
    if (grid > 2048u) grid = 2048u;
    hipLaunchKernelGGL(finalize_batch_kernel, dim3(grid), dim3(bt), 0, s, ws, out_states,
                       out_steps, out_term, out_drift, st);
    return hipGetLastError();
}

hipError_t launch_finalize_frame(const RayWorkspace &ws, const FrameGeom &G, const ShadeParams &S,
                                 int shading, const float *lut, const float *disk_lut, float *out_rgba,
                                 double *out_states, uint32_t *out_steps, uint8_t *out_term,
                                 double *out_drift, FrameStatsDev *st, int n_blocks,
                                 hipStream_t s, uint32_t *wave_cost) {
    if (ws.n == 0) return hipSuccess;
    const size_t lds = (shading && lut)
                           ? (size_t)S.lds_rows * S.lut_w * sizeof(float4) + kDiskLutWidth * sizeof(float)
                           : 0;
    // the attribute belongs to the (function, device) pair: one bit per device, set once
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(finalize_frame_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        if (e != hipSuccess) return e;
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    const uint32_t need = (ws.n + 1023u) / 1024u;
    uint32_t grid = (uint32_t)n_blocks;
    if (grid > need) grid = need;
    if (grid == 0) grid = 1;
    hipLaunchKernelGGL(finalize_frame_kernel, dim3(grid), dim3(1024), lds, s, ws, G, S, shading,
                       reinterpret_cast<const float4 *>(lut), disk_lut, reinterpret_cast<float4 *>(out_rgba),
                       out_states, out_steps, out_term, out_drift, st, wave_cost);
    return hipGetLastError();
}

static uint32_t stream_grid(size_t n) {
    size_t g = (n + kBlock - 1) / kBlock;
    return (uint32_t)(g > 16384 ? 16384 : (g ? g : 1));
}
hipError_t launch_pack_half(const float *rgba, void *half4, size_t n_px, hipStream_t s) {
    if (n_px == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_half_kernel, dim3(stream_grid(n_px)), dim3(kBlock), 0, s,
                       reinterpret_cast<const float4 *>(rgba), static_cast<uint2 *>(half4), n_px);
    return hipGetLastError();
}
hipError_t launch_widen_half(const void *half4, float *rgba, size_t n_px, hipStream_t s) {
    if (n_px == 0) return hipSuccess;
    hipLaunchKernelGGL(widen_half_kernel, dim3(stream_grid(n_px)), dim3(kBlock), 0, s,
                       static_cast<const uint2 *>(half4), reinterpret_cast<float4 *>(rgba), n_px);
    return hipGetLastError();
}
hipError_t launch_unpack_tiles_half(const FrameGeom &G, const void *packed_half4, float *image, hipStream_t s) {
    const size_t total = (size_t)G.n_tiles_local * 4096u;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_tiles_half_kernel, dim3(stream_grid(total)), dim3(kBlock), 0, s, G,
                       static_cast<const uint2 *>(packed_half4), reinterpret_cast<float4 *>(image));
    return hipGetLastError();
}

hipError_t launch_unpack_tiles(const FrameGeom &G, const void *packed, void *image,
                               uint32_t words_per_pixel, hipStream_t s) {
    const size_t total = (size_t)G.n_tiles_local * 4096u * words_per_pixel;
    if (total == 0) return hipSuccess;
    if (words_per_pixel == 4 && (reinterpret_cast<uintptr_t>(packed) & 15u) == 0 &&
        (reinterpret_cast<uintptr_t>(image) & 15u) == 0) { // RGBA f32: one 16-byte move per thread
        size_t g16 = (total / 4 + kBlock - 1) / kBlock;
        if (g16 > 16384) g16 = 16384;
        hipLaunchKernelGGL(unpack_tiles16_kernel, dim3((uint32_t)g16), dim3(kBlock), 0, s, G,
                           static_cast<const uint4 *>(packed), static_cast<uint4 *>(image));
        return hipGetLastError();
    }
    size_t grid = (total + kBlock - 1) / kBlock;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(unpack_tiles_kernel, dim3((uint32_t)grid), dim3(kBlock), 0, s, G,
                       static_cast<const uint32_t *>(packed), static_cast<uint32_t *>(image),
                       words_per_pixel);
    return hipGetLastError();
}

hipError_t launch_wgsl_symplectic(const FrameGeom &G, const WgslParams &P, float *out_rgba,
                                  uint32_t *out_steps, unsigned long long *total_steps,
                                  uint32_t n_slots, hipStream_t s) {
    if (n_slots == 0) return hipSuccess;
    hipLaunchKernelGGL(wgsl_symplectic_kernel, dim3((n_slots + kMarchBlock - 1) / kMarchBlock), dim3(kMarchBlock), 0,
                       s, G, P, reinterpret_cast<float4 *>(out_rgba), out_steps, total_steps, n_slots);
    return hipGetLastError();
}

hipError_t launch_glsl_fragment(const FrameGeom &G, const GlslParams &P, float *out_rgba,
                              uint32_t *out_steps, unsigned long long *total_steps,
                              uint32_t n_slots, hipStream_t s) {
    if (n_slots == 0) return hipSuccess;
    hipLaunchKernelGGL((glsl_fragment_kernel<GRV_ARITH_STRICT>), dim3((n_slots + kMarchBlock - 1) / kMarchBlock), dim3(kMarchBlock), 0, s,
                       G, P, reinterpret_cast<float4 *>(out_rgba), out_steps, total_steps, n_slots, MarchSched{nullptr, nullptr});
    return hipGetLastError();
}

namespace {
// ---- measured-cost dispatch order (engine_types.hpp MarchSched) ----
__global__ __launch_bounds__(kBlock) void march_order_identity_kernel(uint32_t *order, uint32_t *cost, uint32_t n) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        order[i] = i;
        cost[i] = 0u;
    }
}
// One workgroup: counting sort of the blocks by cost, longest first.  1024 buckets of 128 ticks (1.28 us on
// the 100 MHz clock; anything beyond 1.3 ms shares the top bucket).  Whatever the costs are, `order` is a
// permutation of [0, n): every block of the frame is dispatched exactly once.
__global__ __launch_bounds__(1024) void march_rank_kernel(const uint32_t *__restrict__ cost, uint32_t *__restrict__ order,
                                                          uint32_t n) {
    __shared__ uint32_t hist[1024];
    __shared__ uint32_t scan[1024];
    const uint32_t t = threadIdx.x;
    hist[t] = 0u;
    __syncthreads();
    for (uint32_t i = t; i < n; i += 1024u) {
        const uint32_t b = cost[i] >> 7;
        atomicAdd(&hist[b < 1023u ? b : 1023u], 1u);
    }
    __syncthreads();
    // exclusive prefix over the buckets taken from the top: scan[t] = blocks in buckets above t's
    const uint32_t mine = hist[1023u - t];
    scan[t] = mine;
    __syncthreads();
    for (uint32_t off = 1u; off < 1024u; off <<= 1) {
        const uint32_t v = t >= off ? scan[t - off] : 0u;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    hist[1023u - t] = scan[t] - mine; // first position of bucket (1023 - t)
    __syncthreads();
    for (uint32_t i = t; i < n; i += 1024u) {
        const uint32_t b = cost[i] >> 7;
        order[atomicAdd(&hist[b < 1023u ? b : 1023u], 1u)] = i;
    }
}
} // namespace

namespace {
// The compacting schedule's head start (engine.hip run_segments): the sorted wave table of the previous frame, longest
// first, cut at n_head.  Waves order[0 .. n_head) -- the forecast stragglers -- become the slot list `head` (they run in
// ONE unbounded launch beside the chain from the start), the rest the chain's first live list `rest`.
__global__ __launch_bounds__(kBlock) void split_order_kernel(const uint32_t *__restrict__ order, uint32_t n_waves, uint32_t n_head,
                                                            uint32_t *__restrict__ head, uint32_t *__restrict__ rest) {
    const uint32_t total = n_waves * 64u;
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < total; k += gridDim.x * kBlock) {
        const uint32_t w = k >> 6, slot = order[w] * 64u + (k & 63u);
        if (w < n_head) head[k] = slot;
        else rest[k - n_head * 64u] = slot;
    }
}
} // namespace

hipError_t launch_split_order(const uint32_t *order, uint32_t n_waves, uint32_t n_head, uint32_t *head, uint32_t *rest, hipStream_t s) {
    if (n_waves == 0) return hipSuccess;
    uint32_t g = (n_waves * 64u + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(split_order_kernel, dim3(g > 4096u ? 4096u : g), dim3(kBlock), 0, s, order, n_waves, n_head, head, rest);
    return hipGetLastError();
}

hipError_t launch_march_order_identity(uint32_t *order, uint32_t *cost, uint32_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint32_t g = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(march_order_identity_kernel, dim3(g > 1024 ? 1024 : g), dim3(kBlock), 0, s, order, cost, n);
    return hipGetLastError();
}
hipError_t launch_march_rank(const uint32_t *cost, uint32_t *order, uint32_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(march_rank_kernel, dim3(1), dim3(1024), 0, s, cost, order, n);
    return hipGetLastError();
}

namespace {
// evaluates the STRICT unit's sin / cos / pow on the device (grv_strict_math: parity tests
// compare them bit for bit with oracle/ref_libm.c)
__global__ __launch_bounds__(kBlock) void strict_math_kernel(int op, uint32_t n,
                                                            const double *__restrict__ x,
                                                            const double *__restrict__ y,
                                                            double *__restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double s, c;
    if (op & 16) { // f32 forms of the shader-order kernels
        const float xf = (float)x[i], yf = (float)y[i];
        float r;
        switch (op & 15) {
        case 0: case 2: r = sh_sinf(xf); break;
        case 1: case 3: r = sh_cosf(xf); break;
        case 4: r = sh_powf(xf, yf); break;
        case 5: r = sh_expf(xf); break;
        case 7: r = sh_logf(xf); break;
        case 8: r = sh_acosf(xf); break;
        case 9: r = sh_atan2f(xf, yf); break;
        default: r = (float)strictm::sl_atan((double)xf); break;
        }
        out[i] = (double)r;
        return;
    }
    switch (op) {
    case 0: sincos_t<double>(x[i], &s, &c); out[i] = s; break;
    case 1: sincos_t<double>(x[i], &s, &c); out[i] = c; break;
    case 2: out[i] = strictm::sl_sin(x[i]); break;
    case 3: out[i] = strictm::sl_cos(x[i]); break;
    case 4: out[i] = pow_rs(x[i], y[i]); break;
    case 5: out[i] = exp_rs(x[i]); break;
    case 6: out[i] = strictm::sl_atan(x[i]); break;
    case 7: out[i] = strictm::sl_log(x[i]); break;
    case 8: out[i] = strictm::sl_acos(x[i]); break;
    case 10: out[i] = IeeeDiv::div(x[i], IeeeDiv::prep(y[i])); break;
    case 11: out[i] = SharedDiv::div(x[i], SharedDiv::prep(y[i])); break;
    case 12: out[i] = SharedDivNoFixup::div(x[i], SharedDivNoFixup::prep(y[i])); break;
    case 13: out[i] = SharedDiv::prep(x[i]).r; break;
    case 14: {
        const double n = x[i];
        const int d = (int)y[i];
        out[i] = d == 2197 ? ConstDen<2197>::div(n) : d == 216 ? ConstDen<216>::div(n)
               : d == 513 ? ConstDen<513>::div(n) : d == 4104 ? ConstDen<4104>::div(n)
               : d == 27 ? ConstDen<27>::div(n) : d == 2565 ? ConstDen<2565>::div(n)
               : d == 40 ? ConstDen<40>::div(n) : __builtin_nan("");
        break;
    }
    default: out[i] = strictm::sl_atan2(x[i], y[i]); break;
    }
}
} // namespace

namespace {
// the STRICT Kerr-Schild right-hand side at given states, through a chosen division form
// (grv_strict_rhs_probe: the forms must agree bit for bit wherever the guards admit them)
__global__ __launch_bounds__(kBlock) void strict_rhs_probe_kernel(int form, uint32_t n, double M, double a,
                                                                 const double *__restrict__ st,
                                                                 double *__restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Hole<double> bh{M, a, a * a, 2.0 * M};
    bh.divs_ok = divs_ok_hole(M, a);
    bh.divs_nf = divs_nf_hole(M, a);
    const double r = st[8 * i + 1], th = st[8 * i + 2], pt = st[8 * i + 4], pr = st[8 * i + 5],
                 pth = st[8 * i + 6], pph = st[8 * i + 7];
    double sn, cs;
    sincos_t<double>(th, &sn, &cs);
    // per lane here (no ballot): the probe reports which form it was allowed to take
    const bool nf = bh.divs_nf && divs_nf_consts(pt, pph) && divs_nf_point(r, sn, cs);
    const bool ok = bh.divs_ok && divs_ok_point(r, sn, cs);
    Deriv<double> d;
    double used;
    if (form == GRV_RHS_FORM_NOFIXUP && nf) {
        const GInv<double> g = contravariant_ref<GRV_METRIC_KERR_KS, double, SharedDivNoFixup>(bh, r, sn, cs);
        d = rhs_ref_at<GRV_METRIC_KERR_KS, double, SharedDivNoFixup>(bh, r, sn, cs, g, pt, pr, pth, pph);
        used = GRV_RHS_FORM_NOFIXUP;
    } else if (form >= GRV_RHS_FORM_SHARED && ok) {
        const GInv<double> g = contravariant_ref<GRV_METRIC_KERR_KS, double, SharedDiv>(bh, r, sn, cs);
        d = rhs_ref_at<GRV_METRIC_KERR_KS, double, SharedDiv>(bh, r, sn, cs, g, pt, pr, pth, pph);
        used = GRV_RHS_FORM_SHARED;
    } else {
        const GInv<double> g = contravariant_ref<GRV_METRIC_KERR_KS, double>(bh, r, sn, cs);
        d = rhs_ref_at<GRV_METRIC_KERR_KS, double>(bh, r, sn, cs, g, pt, pr, pth, pph);
        used = GRV_RHS_FORM_IEEE;
    }
    double *o = out + 7 * (size_t)i;
    o[0] = d.dt;
    o[1] = d.dr;
    o[2] = d.dth;
    o[3] = d.dph;
    o[4] = d.dpr;
    o[5] = d.dpth;
    o[6] = used;
}
} // namespace

hipError_t launch_strict_rhs_probe(int form, uint32_t n, double M, double a, const double *states, double *out,
                                   hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(strict_rhs_probe_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, form, n, M, a,
                       states, out);
    return hipGetLastError();
}

hipError_t launch_strict_math(int op, uint32_t n, const double *x, const double *y, double *out,
                              hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(strict_math_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, op, n,
                       x, y, out);
    return hipGetLastError();
}

hipError_t launch_spectrum_lut(float *out, uint32_t width, uint32_t height, double max_temp,
                               hipStream_t s) {
    const uint32_t n = width * height;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(spectrum_lut_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s,
                       reinterpret_cast<float4 *>(out), width, height, max_temp);
    return hipGetLastError();
}

namespace {
inline uint32_t at_least_1(uint32_t x) { return x ? x : 1u; }
} // namespace

size_t bloom_scratch_floats(uint32_t w, uint32_t h) {
    const size_t half = (size_t)at_least_1(w / 2) * at_least_1(h / 2);
    const size_t quarter = (size_t)at_least_1(w / 4) * at_least_1(h / 4);
    return 4 * (half + 2 * quarter);
}

hipError_t launch_post_quantize(float *img, uint32_t n_px, hipStream_t s) {
    if (n_px == 0) return hipSuccess;
    hipLaunchKernelGGL(post_quantize_kernel, dim3((n_px + 255u) / 256u), dim3(256), 0, s,
                       reinterpret_cast<float4 *>(img), n_px);
    return hipGetLastError();
}

#define GRV_POST_ARITH GRV_ARITH_STRICT
#define GRV_POST_FN(name) name
#include "post_launch.inc"
#undef GRV_POST_ARITH
#undef GRV_POST_FN

} // namespace grvhip
