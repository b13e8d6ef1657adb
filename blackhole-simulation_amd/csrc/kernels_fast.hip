// kernels_fast.hip -- FAST arithmetic contract (-ffp-contract=fast; reciprocal-based f32 divide / sqrt): the f32
// shader kernels (marches, post chain) and the latency-bound f64 kernels (refill, recorded paths, the one-ray entry).
// The f64 frame kernel of the same contract is kernels_fast_f64.hip (other scheduling flags).
#include "geodesic_kernels.hpp"
#include "wgsl_fast_kernel.hpp"
#include "wgsl_pk_kernel.hpp"
#include <atomic>
#include <cstring>

#include "glsl_fragment.hpp"
#include "post_kernels.hpp"
#include "post_fast_kernels.hpp"

namespace grvhip {

#define GRV_REFILL_ARITH GRV_ARITH_FAST
#define GRV_REFILL_FN launch_refill_fast
#include "refill_launch.inc"
#undef GRV_REFILL_ARITH
#undef GRV_REFILL_FN

#define GRV_PATH_ARITH GRV_ARITH_FAST
#define GRV_PATH_FN launch_path_fast
#include "path_launch.inc"
#undef GRV_PATH_ARITH
#undef GRV_PATH_FN

// grv_integrate_ray_relativistic under the FAST contract (grv_engine_set_ray_arith): the same one-launch
// kernel with the shared-reciprocal right-hand side -- a third of the STRICT instruction count, and a lone
// wave's time is its instruction count
hipError_t launch_single_ray_fast(int kind, const SegmentParams &P, const SingleRayIn &in, double h0,
                                  SingleRayOut *out_pinned, uint32_t seq, hipStream_t s) {
    switch (kind) {
    case GRV_METRIC_KERR_KS:
        hipLaunchKernelGGL((single_ray_kernel<GRV_METRIC_KERR_KS, GRV_ARITH_FAST>), dim3(1), dim3(64), 0, s, P, in, h0, out_pinned, seq);
        break;
    case GRV_METRIC_KERR_BL:
        hipLaunchKernelGGL((single_ray_kernel<GRV_METRIC_KERR_BL, GRV_ARITH_FAST>), dim3(1), dim3(64), 0, s, P, in, h0, out_pinned, seq);
        break;
    case GRV_METRIC_SCHWARZSCHILD:
        hipLaunchKernelGGL((single_ray_kernel<GRV_METRIC_SCHWARZSCHILD, GRV_ARITH_FAST>), dim3(1), dim3(64), 0, s, P, in, h0, out_pinned, seq);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_wgsl_symplectic_fast(const FrameGeom &G, const WgslParams &P, float *out_rgba,
                                       uint32_t *out_steps, unsigned long long *total_steps,
                                       uint32_t n_slots, hipStream_t s) {
    if (n_slots == 0) return hipSuccess;
    hipLaunchKernelGGL(wgsl_symplectic_fast_kernel, dim3((n_slots + kMarchBlock - 1) / kMarchBlock), dim3(kMarchBlock),
                       0, s, G, P, reinterpret_cast<float4 *>(out_rgba), out_steps, total_steps, n_slots);
    return hipGetLastError();
}

// A/B switch: 0 = blocks start in natural (GLSL) / centre-out (WGSL) order, nothing recorded
#ifndef GRV_MARCH_LPT
#define GRV_MARCH_LPT 1
#endif

hipError_t launch_glsl_fragment_fast(const FrameGeom &G, const GlslParams &P, float *out_rgba,
                                     uint32_t *out_steps, unsigned long long *total_steps,
                                     uint32_t n_slots, MarchSched sched, hipStream_t s) {
    if (n_slots == 0) return hipSuccess;
    hipLaunchKernelGGL((glsl_fragment_kernel<GRV_ARITH_FAST>), dim3((n_slots + kMarchBlock - 1) / kMarchBlock),
                       dim3(kMarchBlock), 0, s, G, P, reinterpret_cast<float4 *>(out_rgba), out_steps, total_steps,
                       n_slots, sched);
    return hipGetLastError();
}

hipError_t launch_wgsl_symplectic_pk(const FrameGeom &G, const WgslParams &P, float *out_rgba,
                                     uint32_t *out_steps, unsigned long long *total_steps,
                                     uint32_t n_slots, MarchSched sched, hipStream_t s) {
    if (n_slots == 0) return hipSuccess;
    const uint32_t pairs = (n_slots + 1u) / 2u;
    hipLaunchKernelGGL(wgsl_symplectic_pk_kernel, dim3((pairs + kMarchBlock - 1) / kMarchBlock), dim3(kMarchBlock), 0, s,
                       G, P, reinterpret_cast<float4 *>(out_rgba), out_steps, total_steps, n_slots, sched);
    return hipGetLastError();
}

// entries of MarchSched's arrays for a frame (0: this form takes no measured order).  The long packed march of the
// 8K frame (config 4: 33 M slots, 1 024 steps, a 15 ms launch whose tail is 1 % of it) keeps centre-out: the measured-cost
// order loses 2-3.5 % there at the bench camera (259 200 blocks to sort and look up, profiles/r05_ab_pk_long_cost_order.jsonl)
// and is mixed over the sweep's cameras (-5.7 ... +7.2 %).  Until round 6 the rule read "budget > 512" and so caught every
// 1 024-step march whatever its size: 1080p 205 -> 268 G ray-steps/s with the order, 4K at r0 = 10 M 195 -> 264 G, 4K at the
// bench camera 309 -> 304 G (profiles/r06_ab_pk_order_rule.jsonl).  The rule is the frame's size now.
uint32_t march_blocks_glsl(uint32_t n_slots) { return GRV_MARCH_LPT ? (n_slots + kMarchBlock - 1) / kMarchBlock : 0u; }
uint32_t march_blocks_pk(uint32_t n_slots, int32_t max_steps) {
#ifndef GRV_PK_ORDER_MAX_SLOTS
#define GRV_PK_ORDER_MAX_SLOTS (16u << 20)
#endif
    if (!GRV_MARCH_LPT || (max_steps > 512 && n_slots > GRV_PK_ORDER_MAX_SLOTS)) return 0u;
    return ((n_slots + 1u) / 2u + kMarchBlock - 1) / kMarchBlock;
}

#ifdef GRV_MARCH_TIMELINE
// instrument hook of an A/B library only (never in the product build): where glsl_fragment_kernel<FAST>
// writes its per-wave records; nullptr switches them off
extern "C" int grv_debug_set_march_timeline(unsigned long long *d_buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_march_timeline), &d_buf, sizeof d_buf);
}
#endif

namespace {
inline uint32_t at_least_1(uint32_t x) { return x ? x : 1u; }
} // namespace
#define GRV_POST_ARITH GRV_ARITH_FAST
#define GRV_POST_FN(name) name##_fast
#define GRV_POST_FAST_FORMS
#include "post_launch.inc"
#undef GRV_POST_FAST_FORMS
#undef GRV_POST_ARITH
#undef GRV_POST_FN

} // namespace grvhip
