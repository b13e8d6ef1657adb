// kernels_fast_f64.hip -- FAST arithmetic contract of the f64 geodesic kernels (-ffp-contract=fast): the segment,
// refill, path and one-ray kernels with the shared-reciprocal Kerr-Schild right-hand side (kerr_device.hpp:
// rhs_ks_geom) and FMA contraction.  Its own translation unit because it is compiled WITHOUT the post-RA machine
// scheduler (-mllvm -enable-post-misched=0, Makefile): the RKF45 try loop issues 1.6 % faster in the order the
// pre-RA scheduler leaves (profiles/r05_ab_f64_post_ra_sched.jsonl) -- same instructions, same results.
#include "geodesic_kernels.hpp"
#include <atomic>
#include <cstring>

namespace grvhip {

namespace {
template <int KIND, int METHOD>
hipError_t go(const RayWorkspace &ws, const SegmentParams &P, const uint32_t *live_in,
              uint32_t n_live, uint32_t *live_out, uint32_t *live_out_count, hipStream_t s) {
    const uint32_t grid = (n_live + kSegBlock - 1) / kSegBlock;
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL((integrate_segment_kernel<KIND, GRV_ARITH_FAST, METHOD>), dim3(grid),
                       dim3(kSegBlock), 0, s, ws, P, live_in, n_live, live_out, live_out_count);
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

hipError_t launch_segment_fast(int kind, int method, const RayWorkspace &ws,
                               const SegmentParams &P, const uint32_t *live_in, uint32_t n_live,
                               uint32_t *live_out, uint32_t *live_out_count, hipStream_t s) {
    switch (kind) {
    case GRV_METRIC_KERR_KS: return by_method<GRV_METRIC_KERR_KS>(method, ws, P, live_in, n_live, live_out, live_out_count, s);
    case GRV_METRIC_KERR_BL: return by_method<GRV_METRIC_KERR_BL>(method, ws, P, live_in, n_live, live_out, live_out_count, s);
    case GRV_METRIC_SCHWARZSCHILD: return by_method<GRV_METRIC_SCHWARZSCHILD>(method, ws, P, live_in, n_live, live_out, live_out_count, s);
    default: return hipErrorInvalidValue;
    }
}

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

} // namespace grvhip
