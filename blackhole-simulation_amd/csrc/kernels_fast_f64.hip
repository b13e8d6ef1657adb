// kernels_fast_f64.hip -- the frame kernel of the FAST f64 contract (-ffp-contract=fast): integrate_segment_kernel
// with the shared-reciprocal Kerr-Schild right-hand side (kerr_device.hpp: rhs_ks_geom) and FMA contraction.  Its own
// translation unit because it is compiled WITHOUT the post-RA machine scheduler (-mllvm -enable-post-misched=0,
// Makefile): at three full waves per SIMD the RKF45 try loop runs 1.6-2.1 % faster in the order the pre-RA scheduler
// leaves (profiles/r05_ab_f64_post_ra_sched.jsonl) -- same instructions, same results.  The latency-bound FAST f64
// kernels (refill for small batches, recorded paths, the one-ray entry: a lone wave wants the interleaving the
// post-RA pass provides, 373 against 398 us on the doc-test ray) stay in kernels_fast.hip.
#include "geodesic_kernels.hpp"

namespace grvhip {

namespace {
template <int KIND, int METHOD>
hipError_t go(const RayWorkspace &ws, const SegmentParams &P, const uint32_t *live_in,
              uint32_t n_live, uint32_t *live_out, uint32_t *live_out_count, hipStream_t s) {
    const uint32_t threads = segment_block_threads(live_out, n_live, P.order);
    const uint32_t grid = (n_live + threads - 1) / threads;
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL((integrate_segment_kernel<KIND, GRV_ARITH_FAST, METHOD>), dim3(grid),
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

#define GRV_COMPACT_ARITH GRV_ARITH_FAST
#define GRV_COMPACT_FN launch_compact_fast
#include "compact_launch.inc"
#undef GRV_COMPACT_ARITH
#undef GRV_COMPACT_FN

} // namespace grvhip
