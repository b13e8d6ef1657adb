// geodesic_kernels.hpp -- wavefront geodesic integrator kernels for gfx950.
//
// Execution model (one "segment" launch):
//   * per-ray state lives in HBM as struct-of-arrays (RayWorkspace): lane i of a
//     wave touches element i of each array -> every load/store is one coalesced
//     512-byte transaction per f64 component;
//   * a launch loads the state of the live rays listed in `live_in`, runs up to
//     `max_tries` integrator tries per ray entirely in registers, stores it back;
//   * a wave leaves the loop as soon as __ballot(live) == 0 (adaptive-step early out);
//   * surviving rays are appended to `live_out` with one wave-aggregated atomic
//     (ballot + popcount + mbcnt prefix): this is the ray compaction between launches.
//
// Reference behaviour restated (physics-engine/gravitas-core/src):
//   integrate()               geodesic/mod.rs:180-253
//   check_termination         geodesic/mod.rs:256-265
//   AdaptiveStepper::step     geodesic/integrator.rs:72-107 (constants :61-68)
//   adaptive_rkf45_step       geodesic/integrator.rs:113-190
//   step_rk4 / step_symplectic geodesic/integrator.rs:193-226
// Compiled twice (STRICT: -ffp-contract=off, FAST: -ffp-contract=fast); GRV_TU_ARITH
// selects which arithmetic contract this translation unit instantiates.
#pragma once

#include "engine_types.hpp"
#include "kerr_device.hpp"

namespace {

using namespace grvhip;

// ---------------------------------------------------------------------------
// register-resident ray
// ---------------------------------------------------------------------------
struct RayRegs {
    double t, r, th, ph, pr, pth;
    double pt, pph;
    double h;
    double drift;
    uint32_t steps, tries, flags;
    uint32_t phase; // steps % renorm_interval, carried incrementally (no division in the loop)
    bool nf_ok; // STRICT Kerr-Schild: M, a, p_t, p_phi admit the NOFIX forms (ray_resume)
    Deriv<double> k1; // FAST Kerr-Schild only: right-hand side at the current state, formed by
                      // the post-step bookkeeping and reused as stage 1 of the next try
};

// The right-hand side at the current state is formed once, by the post-step bookkeeping, and reused
// as stage 1 of the next try (also across rejected tries).  FAST Kerr-Schild builds it from the
// shared geometry; the STRICT forms evaluate rhs_ref_at on the sine, cosine and inverse metric the
// bookkeeping computed at that point -- the very operations stage 1 would repeat, so the bits are
// those of the uncached form (GRV_STRICT_STAGE1_CACHE=0 compiles that form for comparison).
#ifndef GRV_STRICT_STAGE1_CACHE
#define GRV_STRICT_STAGE1_CACHE 1
#endif
template <int KIND, int ARITH> constexpr bool kFastKsCache = (KIND == GRV_METRIC_KERR_KS && ARITH == GRV_ARITH_FAST);
template <int KIND, int ARITH> constexpr bool kStrictCache = (ARITH == GRV_ARITH_STRICT && GRV_STRICT_STAGE1_CACHE != 0);
template <int KIND, int ARITH> constexpr bool kStage1Cache = kFastKsCache<KIND, ARITH> || kStrictCache<KIND, ARITH>;

__device__ __forceinline__ Hole<double> make_hole(const SegmentParams &P) {
    Hole<double> bh{P.M, P.a, P.a2, 2.0 * P.M};
    bh.divs_ok = divs_ok_hole(P.M, P.a);
    bh.divs_nf = divs_nf_hole(P.M, P.a);
    return bh;
}

__device__ __forceinline__ double clamp_rs(double x, double lo, double hi) {
    // Rust f64::clamp
    return x < lo ? lo : (x > hi ? hi : x);
}
__device__ __forceinline__ double signum_rs(double x) {
    return (x != x) ? x : (signbit(x) ? -1.0 : 1.0);
}

// One Fehlberg 4(5) evaluation on the reduced state.  Writes the 5th-order weighted
// stage sums into `inc` (the candidate is y + h * inc, formed by the caller only if the
// try is accepted, in place) and returns the error estimate (max-abs over t,r,theta,phi).
template <int KIND, int ARITH>
__device__ __forceinline__ double rkf45_try(const Hole<double> &bh, const RayRegs &y, double h,
                                            Deriv<double> &inc, const KsRayConsts &rc) {
    // right-hand side at a stage point; the FAST Kerr-Schild form takes the per-ray
    // constant products (rc) instead of recomputing them six times a try
    auto f = [&](double r_, double th_, double pr_, double pth_) {
        if constexpr (kFastKsCache<KIND, ARITH>)
            return rhs_ks_geom(bh, ks_geom(bh, r_, th_), r_, rc, pr_, pth_);
        else
            return rhs<KIND, ARITH>(bh, r_, th_, y.pt, pr_, pth_, y.pph, y.nf_ok);
    };
    // stage scale factors, same expressions as integrator.rs:119-160
    double s21, s31, s32, s41, s42, s43, s51, s52, s53, s54, s61, s62, s63, s64, s65;
    if constexpr (ARITH == GRV_ARITH_STRICT) {
        s21 = h / 4.0;
        s31 = 3.0 * h / 32.0;
        s32 = 9.0 * h / 32.0;
        s52 = -8.0 * h;
        s62 = 2.0 * h;
        // the ten quotients by a constant that is not a power of two: the same IEEE quotients through
        // the literal-reciprocal form (ConstDen) when the whole wave's step sizes admit it -- always,
        // after the first try: the controller keeps 1e-5 <= |h| <= 10
        if (GRV_STRICT_CONSTDEN && __ballot(!const_div_ok(h)) == 0ull) {
            s41 = ConstDen<2197>::div(1932.0 * h);
            s42 = ConstDen<2197>::div(-7200.0 * h);
            s43 = ConstDen<2197>::div(7296.0 * h);
            s51 = ConstDen<216>::div(439.0 * h);
            s53 = ConstDen<513>::div(3680.0 * h);
            s54 = ConstDen<4104>::div(-845.0 * h);
            s61 = ConstDen<27>::div(-8.0 * h);
            s63 = ConstDen<2565>::div(-3544.0 * h);
            s64 = ConstDen<4104>::div(1859.0 * h);
            s65 = ConstDen<40>::div(-11.0 * h);
        } else {
            s41 = 1932.0 * h / 2197.0;
            s42 = -7200.0 * h / 2197.0;
            s43 = 7296.0 * h / 2197.0;
            s51 = 439.0 * h / 216.0;
            s53 = 3680.0 * h / 513.0;
            s54 = -845.0 * h / 4104.0;
            s61 = -8.0 * h / 27.0;
            s63 = -3544.0 * h / 2565.0;
            s64 = 1859.0 * h / 4104.0;
            s65 = -11.0 * h / 40.0;
        }
    } else {
        s21 = h * 0.25;
        s31 = h * (3.0 / 32.0);
        s32 = h * (9.0 / 32.0);
        s41 = h * (1932.0 / 2197.0);
        s42 = h * (-7200.0 / 2197.0);
        s43 = h * (7296.0 / 2197.0);
        s51 = h * (439.0 / 216.0);
        s52 = h * -8.0;
        s53 = h * (3680.0 / 513.0);
        s54 = h * (-845.0 / 4104.0);
        s61 = h * (-8.0 / 27.0);
        s62 = h * 2.0;
        s63 = h * (-3544.0 / 2565.0);
        s64 = h * (1859.0 / 4104.0);
        s65 = h * (-11.0 / 40.0);
    }
    constexpr double c1 = 16.0 / 135.0, c3 = 6656.0 / 12825.0, c4 = 28561.0 / 56430.0,
                     c5 = 9.0 / 50.0, c6 = 2.0 / 55.0;
    constexpr double e1 = 16.0 / 135.0 - 25.0 / 216.0, e3 = 6656.0 / 12825.0 - 1408.0 / 2565.0,
                     e4 = 28561.0 / 56430.0 - 2197.0 / 4104.0, e5 = -9.0 / 50.0 + 1.0 / 5.0;

    Deriv<double> k1;
    if constexpr (kStage1Cache<KIND, ARITH>)
        k1 = y.k1;
    else
        k1 = rhs<KIND, ARITH>(bh, y.r, y.th, y.pt, y.pr, y.pth, y.pph, y.nf_ok);
    // t and phi never feed back into the right-hand side: keep only their running
    // 5th-order and error sums (same left-to-right order as the reference).
    double a5_t = c1 * k1.dt, a5_ph = c1 * k1.dph;
    double ae_t = e1 * k1.dt, ae_ph = e1 * k1.dph;

    // FAST: nested-fma stage sums (one instruction per term); STRICT keeps the reference's
    // (k1*s1 + k2*s2 + ...) grouping.
    auto st2 = [&](double y0, double a1, double a2) {
        if constexpr (ARITH == GRV_ARITH_FAST) return fma(a2, s32, fma(a1, s31, y0));
        else return y0 + (a1 * s31 + a2 * s32);
    };
    auto st3 = [&](double y0, double a1, double a2, double a3) {
        if constexpr (ARITH == GRV_ARITH_FAST) return fma(a3, s43, fma(a2, s42, fma(a1, s41, y0)));
        else return y0 + (a1 * s41 + a2 * s42 + a3 * s43);
    };
    auto st4 = [&](double y0, double a1, double a2, double a3, double a4) {
        if constexpr (ARITH == GRV_ARITH_FAST)
            return fma(a4, s54, fma(a3, s53, fma(a2, s52, fma(a1, s51, y0))));
        else return y0 + (a1 * s51 + a2 * s52 + a3 * s53 + a4 * s54);
    };
    auto st5 = [&](double y0, double a1, double a2, double a3, double a4, double a5) {
        if constexpr (ARITH == GRV_ARITH_FAST)
            return fma(a5, s65, fma(a4, s64, fma(a3, s63, fma(a2, s62, fma(a1, s61, y0)))));
        else return y0 + (a1 * s61 + a2 * s62 + a3 * s63 + a4 * s64 + a5 * s65);
    };

    const Deriv<double> k2 =
        f(y.r + k1.dr * s21, y.th + k1.dth * s21, y.pr + k1.dpr * s21, y.pth + k1.dpth * s21);

    const Deriv<double> k3 = f(st2(y.r, k1.dr, k2.dr), st2(y.th, k1.dth, k2.dth),
                               st2(y.pr, k1.dpr, k2.dpr), st2(y.pth, k1.dpth, k2.dpth));
    a5_t = a5_t + c3 * k3.dt;
    a5_ph = a5_ph + c3 * k3.dph;
    ae_t = ae_t + e3 * k3.dt;
    ae_ph = ae_ph + e3 * k3.dph;

    const Deriv<double> k4 =
        f(st3(y.r, k1.dr, k2.dr, k3.dr), st3(y.th, k1.dth, k2.dth, k3.dth),
          st3(y.pr, k1.dpr, k2.dpr, k3.dpr), st3(y.pth, k1.dpth, k2.dpth, k3.dpth));
    a5_t = a5_t + c4 * k4.dt;
    a5_ph = a5_ph + c4 * k4.dph;
    ae_t = ae_t + e4 * k4.dt;
    ae_ph = ae_ph + e4 * k4.dph;

    const Deriv<double> k5 =
        f(st4(y.r, k1.dr, k2.dr, k3.dr, k4.dr), st4(y.th, k1.dth, k2.dth, k3.dth, k4.dth),
          st4(y.pr, k1.dpr, k2.dpr, k3.dpr, k4.dpr), st4(y.pth, k1.dpth, k2.dpth, k3.dpth, k4.dpth));
    a5_t = a5_t - c5 * k5.dt;
    a5_ph = a5_ph - c5 * k5.dph;
    ae_t = ae_t + e5 * k5.dt;
    ae_ph = ae_ph + e5 * k5.dph;

    const Deriv<double> k6 = f(st5(y.r, k1.dr, k2.dr, k3.dr, k4.dr, k5.dr),
                               st5(y.th, k1.dth, k2.dth, k3.dth, k4.dth, k5.dth),
                               st5(y.pr, k1.dpr, k2.dpr, k3.dpr, k4.dpr, k5.dpr),
                               st5(y.pth, k1.dpth, k2.dpth, k3.dpth, k4.dpth, k5.dpth));
    a5_t = a5_t + c6 * k6.dt;
    a5_ph = a5_ph + c6 * k6.dph;
    ae_t = ae_t + c6 * k6.dt;
    ae_ph = ae_ph + c6 * k6.dph;

    inc.dt = a5_t;
    inc.dph = a5_ph;
    inc.dr = c1 * k1.dr + c3 * k3.dr + c4 * k4.dr - c5 * k5.dr + c6 * k6.dr;
    inc.dth = c1 * k1.dth + c3 * k3.dth + c4 * k4.dth - c5 * k5.dth + c6 * k6.dth;
    inc.dpr = c1 * k1.dpr + c3 * k3.dpr + c4 * k4.dpr - c5 * k5.dpr + c6 * k6.dpr;
    inc.dpth = c1 * k1.dpth + c3 * k3.dpth + c4 * k4.dpth - c5 * k5.dpth + c6 * k6.dpth;

    const double err_r = h * (e1 * k1.dr + e3 * k3.dr + e4 * k4.dr + e5 * k5.dr + c6 * k6.dr);
    const double err_th = h * (e1 * k1.dth + e3 * k3.dth + e4 * k4.dth + e5 * k5.dth + c6 * k6.dth);
    double error = fmax(0.0, fabs(h * ae_t));
    error = fmax(error, fabs(err_r));
    error = fmax(error, fabs(err_th));
    error = fmax(error, fabs(h * ae_ph));
    return error;
}

template <int KIND, int ARITH>
__device__ __forceinline__ void rk4_step(const Hole<double> &bh, RayRegs &y, double h) {
    const double hh = 0.5 * h;
    Deriv<double> k1;
    if constexpr (kStage1Cache<KIND, ARITH>)
        k1 = y.k1; // the right-hand side at the current state (post-step bookkeeping)
    else
        k1 = rhs<KIND, ARITH>(bh, y.r, y.th, y.pt, y.pr, y.pth, y.pph, y.nf_ok);
    const Deriv<double> k2 = rhs<KIND, ARITH>(bh, y.r + k1.dr * hh, y.th + k1.dth * hh, y.pt,
                                              y.pr + k1.dpr * hh, y.pth + k1.dpth * hh, y.pph, y.nf_ok);
    const Deriv<double> k3 = rhs<KIND, ARITH>(bh, y.r + k2.dr * hh, y.th + k2.dth * hh, y.pt,
                                              y.pr + k2.dpr * hh, y.pth + k2.dpth * hh, y.pph, y.nf_ok);
    const Deriv<double> k4 = rhs<KIND, ARITH>(bh, y.r + k3.dr * h, y.th + k3.dth * h, y.pt,
                                              y.pr + k3.dpr * h, y.pth + k3.dpth * h, y.pph, y.nf_ok);
    const double h6 = h / 6.0;
    y.t += h6 * (k1.dt + 2.0 * k2.dt + 2.0 * k3.dt + k4.dt);
    y.r += h6 * (k1.dr + 2.0 * k2.dr + 2.0 * k3.dr + k4.dr);
    y.th += h6 * (k1.dth + 2.0 * k2.dth + 2.0 * k3.dth + k4.dth);
    y.ph += h6 * (k1.dph + 2.0 * k2.dph + 2.0 * k3.dph + k4.dph);
    y.pr += h6 * (k1.dpr + 2.0 * k2.dpr + 2.0 * k3.dpr + k4.dpr);
    y.pth += h6 * (k1.dpth + 2.0 * k2.dpth + 2.0 * k3.dpth + k4.dpth);
}

template <int KIND, int ARITH>
__device__ __forceinline__ void symplectic_step(const Hole<double> &bh, RayRegs &y, double h) {
    // implicit midpoint, exactly two fixed-point sweeps (integrator.rs:209-226).
    // t and phi of the midpoint never enter the right-hand side.
    double mr = y.r, mth = y.th, mpr = y.pr, mpth = y.pth;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        Deriv<double> d;
        if constexpr (kStage1Cache<KIND, ARITH>) {
            // the first sweep evaluates the right-hand side at the current state: cached
            if (it == 0) d = y.k1;
            else d = rhs<KIND, ARITH>(bh, mr, mth, y.pt, mpr, mpth, y.pph, y.nf_ok);
        } else {
            d = rhs<KIND, ARITH>(bh, mr, mth, y.pt, mpr, mpth, y.pph, y.nf_ok);
        }
        mr = 0.5 * (y.r + (y.r + d.dr * h));
        mth = 0.5 * (y.th + (y.th + d.dth * h));
        mpr = 0.5 * (y.pr + (y.pr + d.dpr * h));
        mpth = 0.5 * (y.pth + (y.pth + d.dpth * h));
    }
    const Deriv<double> f = rhs<KIND, ARITH>(bh, mr, mth, y.pt, mpr, mpth, y.pph, y.nf_ok);
    y.t += f.dt * h;
    y.r += f.dr * h;
    y.th += f.dth * h;
    y.ph += f.dph * h;
    y.pr += f.dpr * h;
    y.pth += f.dpth * h;
}

__device__ __forceinline__ uint32_t termination_of(double r, const SegmentParams &P) {
    if (r < P.horizon_limit) return GRV_TERM_HORIZON;
    if (r > P.escape_radius) return GRV_TERM_ESCAPE;
    return GRV_TERM_NONE;
}

// A live ray has flags & kFlagTermMask == NONE and steps < max_steps.
__device__ __forceinline__ bool ray_live(const RayRegs &y) {
    return (y.flags & (kFlagTermMask | kFlagValid)) == kFlagValid;
}

// Start-of-trajectory bookkeeping: mod.rs:200 (initial renormalisation), the
// clamp of AdaptiveStepper::step's first h_try, and the first loop-top checks.
template <int KIND, int ARITH = GRV_ARITH_STRICT>
__device__ __forceinline__ void ray_begin(const Hole<double> &bh, RayRegs &y,
                                          const SegmentParams &P, bool adaptive) {
    const GInv<double> g = contravariant_at<KIND, ARITH, double>(bh, y.r, y.th);
    y.pr = renormalized_pr<KIND, ARITH, double>(g, y.pt, y.pr, y.pth, y.pph);
    if (adaptive) y.h = clamp_rs(y.h, -10.0, 10.0);
    uint32_t term = GRV_TERM_NONE;
    if (P.max_steps == 0)
        term = GRV_TERM_MAXSTEPS;
    else
        term = termination_of(y.r, P);
    y.flags = (y.flags & ~kFlagTermMask) | term;
}

// STRICT post-step metric work at the new point: projection (on the renormalisation phase), H,
// and the right-hand side there for the stage-1 cache.  Returns H.
template <int KIND>
__device__ __forceinline__ double post_step_ref(const Hole<double> &bh, RayRegs &y, bool renorm) {
    double s, c;
    sincos_t(y.th, &s, &c);
    auto body = [&](auto div) {
        using DIV = decltype(div);
        const GInv<double> g = contravariant_ref<KIND, double, DIV>(bh, y.r, s, c);
        if (renorm) y.pr = renormalized_pr<KIND, GRV_ARITH_STRICT, double, DIV::kNoFixup>(g, y.pt, y.pr, y.pth, y.pph);
        const double hv = hamiltonian_of<KIND, double, DIV::kNoFixup>(g, y.pt, y.pr, y.pth, y.pph);
        y.k1 = rhs_ref_at<KIND, double, DIV>(bh, y.r, s, c, g, y.pt, y.pr, y.pth, y.pph);
        return hv;
    };
    if constexpr (KIND == GRV_METRIC_KERR_KS) {
        const bool nf = GRV_STRICT_NOFIXUP && y.nf_ok && divs_nf_point(y.r, s, c);
        if (GRV_STRICT_NOFIXUP && __ballot(!nf) == 0ull) return body(SharedDivNoFixup{});
        const bool ok = bh.divs_ok && divs_ok_point(y.r, s, c);
        if (__ballot(!ok) == 0ull) return body(SharedDiv{});
    }
    return body(IeeeDiv{});
}

// Everything integrate() does after a completed step (mod.rs:228-239), then the
// next iteration's loop-top checks, plus the disk-plane crossing recorder.
template <int KIND, int ARITH>
__device__ __forceinline__ void after_step(const Hole<double> &bh, RayRegs &y, double r_prev,
                                           double th_prev, const SegmentParams &P,
                                           const RayWorkspace &ws, uint32_t slot,
                                           const KsRayConsts &rc) {
    const bool renorm = P.renorm_interval != 0 && y.phase == 0u; // steps % interval == 0
    double hv;
    if constexpr (kFastKsCache<KIND, ARITH>) {
        // one geometry evaluation serves the projection, H and stage 1 of the next try
        const KsGeom geom = ks_geom(bh, y.r, y.th);
        if (renorm)
            y.pr = renormalized_pr<KIND, ARITH, double>(ginv_from_geom(bh, geom, y.r), y.pt, y.pr,
                                                        y.pth, y.pph);
        y.k1 = rhs_ks_geom(bh, geom, y.r, rc, y.pr, y.pth, &hv);
        hv = fabs(hv);
    } else if constexpr (kStrictCache<KIND, ARITH>) {
        // reference order: contravariant() for the projection and for H (mod.rs:229-237), then the
        // next iteration's first derivative at the same point -- one sine / cosine / metric for all
        hv = fabs(post_step_ref<KIND>(bh, y, renorm));
    } else {
        const GInv<double> g = contravariant_at<KIND, ARITH, double>(bh, y.r, y.th);
        if (renorm) y.pr = renormalized_pr<KIND, ARITH, double>(g, y.pt, y.pr, y.pth, y.pph);
        hv = fabs(hamiltonian_of<KIND, double>(g, y.pt, y.pr, y.pth, y.pph));
    }
    if (hv > y.drift) y.drift = hv;
    y.steps += 1;
    y.phase = (y.phase + 1u == P.renorm_interval) ? 0u : y.phase + 1u;

    uint32_t term = GRV_TERM_NONE;
    if (P.shading) {
        constexpr double kPi2 = 1.57079632679489661923;
        if ((th_prev - kPi2) * (y.th - kPi2) <= 0.0) {
            const double dth = y.th - th_prev;
            const double f = (dth == 0.0) ? 0.0 : (kPi2 - th_prev) / dth;
            const double r_c = r_prev + f * (y.r - r_prev);
            if (r_c > P.disk_inner && r_c < P.disk_outer) {
                uint32_t nc = (y.flags & kFlagCrossMask) >> kFlagCrossShift;
                if (nc < (uint32_t)kMaxCrossRec) ws.rc[(size_t)nc * ws.n + slot] = r_c;
                nc = nc < 15u ? nc + 1u : nc;
                y.flags = (y.flags & ~kFlagCrossMask) | (nc << kFlagCrossShift);
                if (nc >= P.max_crossings) term = GRV_TERM_DISK_CROSSING;
            }
        }
    }
    if (term == GRV_TERM_NONE) {
        if (y.steps >= P.max_steps)
            term = GRV_TERM_MAXSTEPS;
        else
            term = termination_of(y.r, P);
    }
    y.flags = (y.flags & ~kFlagTermMask) | term;
}

__device__ __forceinline__ void load_ray(const RayWorkspace &ws, uint32_t s, RayRegs &y) {
    y.t = ws.t[s];
    y.r = ws.r[s];
    y.th = ws.th[s];
    y.ph = ws.ph[s];
    y.pr = ws.pr[s];
    y.pth = ws.pth[s];
    y.pt = ws.pt[s];
    y.pph = ws.pph[s];
    y.h = ws.h[s];
    y.drift = ws.drift[s];
    y.steps = ws.steps[s];
    y.tries = ws.tries[s];
    y.flags = ws.flags[s];
}

__device__ __forceinline__ void store_ray(const RayWorkspace &ws, uint32_t s, const RayRegs &y) {
    ws.t[s] = y.t;
    ws.r[s] = y.r;
    ws.th[s] = y.th;
    ws.ph[s] = y.ph;
    ws.pr[s] = y.pr;
    ws.pth[s] = y.pth;
    ws.h[s] = y.h;
    ws.drift[s] = y.drift;
    ws.steps[s] = y.steps;
    ws.tries[s] = y.tries;
    ws.flags[s] = y.flags;
}

// ---------------------------------------------------------------------------
// One integrator try of a live ray (integrator.rs:83-123 for RKF45) and, when it
// completes a step, the loop-body epilogue.  Returns the ray's liveness.
// ---------------------------------------------------------------------------
template <int KIND, int ARITH, int METHOD>
__device__ __forceinline__ bool advance_one(const Hole<double> &bh, RayRegs &y,
                                            const SegmentParams &P, const RayWorkspace &ws,
                                            uint32_t slot, const KsRayConsts &rc) {
    const double r_prev = y.r, th_prev = y.th;
    bool stepped;
    if constexpr (METHOD == GRV_METHOD_RKF45) {
        Deriv<double> inc;
        const double h = y.h;
        const double err = rkf45_try<KIND, ARITH>(bh, y, h, inc, rc);
        y.tries += 1;
        const bool forced = (y.flags & kFlagForced) != 0u;
        double ratio;
        if constexpr (ARITH == GRV_ARITH_FAST)
            ratio = err * P.inv_tolerance; // err == 0 -> 0 without the special case
        else
            ratio = (err == 0.0) ? 0.0 : err / P.tolerance;
        // integrator.rs:99-104: the forced minimum step is taken unconditionally and
        // its own size is handed back as the next h.
        stepped = forced || ratio <= 1.0;
        if (stepped) {
            y.t = y.t + h * inc.dt;
            y.r = y.r + h * inc.dr;
            y.th = y.th + h * inc.dth;
            y.ph = y.ph + h * inc.dph;
            y.pr = y.pr + h * inc.dpr;
            y.pth = y.pth + h * inc.dpth;
            if (!forced) {
                double growth;
                if constexpr (ARITH == GRV_ARITH_FAST)
                    growth = (ratio < 1e-4) ? 5.0 : 0.9 * fast_pow_m1_5(ratio);
                else
                    growth = (ratio < 1e-4) ? 5.0 : 0.9 * pow_rs(ratio, -0.2);
                y.h = clamp_rs(h * fmin(growth, 5.0), -10.0, 10.0);
            }
            y.flags &= ~kFlagForced;
        } else {
            double shrink;
            if constexpr (ARITH == GRV_ARITH_FAST)
                shrink = 0.9 * fast_pow_m1_4(ratio);
            else
                shrink = 0.9 * pow_rs(ratio, -0.25);
            double hn = h * fmax(shrink, 0.1);
            if (fabs(hn) < 1e-5) {
                hn = 1e-5 * signum_rs(hn);
                y.flags |= kFlagForced;
            }
            y.h = hn;
        }
    } else if constexpr (METHOD == GRV_METHOD_RK4) {
        rk4_step<KIND, ARITH>(bh, y, P.step_size);
        y.tries += 1;
        stepped = true;
    } else {
        symplectic_step<KIND, ARITH>(bh, y, P.step_size);
        y.tries += 1;
        stepped = true;
    }
    if (stepped) {
        after_step<KIND, ARITH>(bh, y, r_prev, th_prev, P, ws, slot, rc);
        return ray_live(y);
    }
    return true;
}

// Per-ray values derived from a state that just came from HBM.
template <int KIND, int ARITH>
__device__ __forceinline__ void ray_resume(const Hole<double> &bh, RayRegs &y,
                                           const SegmentParams &P, bool live, KsRayConsts &rc) {
    y.phase = P.renorm_interval ? y.steps % P.renorm_interval : 1u;
    rc = ks_ray_consts(bh, y.pt, y.pph); // p_t, p_phi never change
    y.nf_ok = bh.divs_nf && divs_nf_consts(y.pt, y.pph);
    if constexpr (kFastKsCache<KIND, ARITH>) {
        // rebuild the stage-1 cache
        if (live) y.k1 = rhs_ks_geom(bh, ks_geom(bh, y.r, y.th), y.r, rc, y.pr, y.pth);
    } else if constexpr (kStrictCache<KIND, ARITH>) {
        if (live) y.k1 = rhs<KIND, ARITH>(bh, y.r, y.th, y.pt, y.pr, y.pth, y.pph, y.nf_ok);
    }
}

// ---------------------------------------------------------------------------
// The segment kernel.
// ---------------------------------------------------------------------------
// waves per SIMD the segment kernel is compiled for at least: the Kerr-Schild RKF45 forms sit at the
// three-wave boundary (code objects of this tree: FAST 152 VGPRs, STRICT 168 -- the attribute keeps both at three
// waves; tools/kernel_resources.py, tests/test_code_objects.py)
template <int KIND, int METHOD>
constexpr int kSegmentWavesMin = (KIND == GRV_METRIC_KERR_KS && METHOD == GRV_METHOD_RKF45) ? 3 : 1;

#ifndef GRV_WS_RELOAD
#define GRV_WS_RELOAD 1
#endif
// Threads per block of the segment kernel: a property of the launch (blockDim.x, read where it is needed), not of
// the code.  The one-launch schedule starts one-wave blocks -- a finished wave's slot goes back to the dispatcher at
// once instead of waiting for the slowest of four: c3 +1.1 %, c5 +1.1 % (profiles/r05_ab_segment_block.jsonl) --,
// the compacting schedule four-wave blocks (one atomic per block on the live-list counter: one-wave blocks
// cost it 2 %).  kSegBlock is the larger of the two (launch bound, LDS of the append).
#ifndef GRV_SEGMENT_BLOCK
#define GRV_SEGMENT_BLOCK 256
#endif
#ifndef GRV_SEGMENT_BLOCK_ONE_LAUNCH
#define GRV_SEGMENT_BLOCK_ONE_LAUNCH 64
#endif
constexpr int kSegBlock = GRV_SEGMENT_BLOCK;
constexpr int kSegBlockOneLaunch = GRV_SEGMENT_BLOCK_ONE_LAUNCH;
// the launch shape of a segment launch: no live list to append to = the one-launch schedule.  Small launches
// (a rank's eighth of the 4K frame, 1.04 M rays, with two frames in flight: 3.56 against 3.64 ms) still prefer
// four-wave blocks; from 2 M rays on one-wave blocks are level or ahead (profiles/r05_rank_share_segment_block.txt)
// (kSegOneWaveMinRays: engine_types.hpp)
// A launch that is handed a dispatch order starts one-wave blocks whatever its size: the order's entries are one-wave blocks.
__host__ inline uint32_t segment_block_threads(const uint32_t *live_out, uint32_t n_live, const uint32_t *order) {
    if (live_out) return (uint32_t)kSegBlock;
    return (order || n_live >= kSegOneWaveMinRays) ? (uint32_t)kSegBlockOneLaunch : (uint32_t)kSegBlock;
}

template <int KIND, int ARITH, int METHOD>
__global__ __launch_bounds__(kSegBlock) __attribute__((amdgpu_waves_per_eu(kSegmentWavesMin<KIND, METHOD>)))
void integrate_segment_kernel(
    RayWorkspace ws, SegmentParams P, const uint32_t *__restrict__ live_in, uint32_t n_live,
    uint32_t *__restrict__ live_out, uint32_t *__restrict__ live_out_count) {
    const uint32_t blk = P.order ? P.order[blockIdx.x] : dispatch_block_f64(blockIdx.x, gridDim.x, P.block_order);
    const uint32_t k = blk * blockDim.x + threadIdx.x;
    const bool have = k < n_live;
    const uint32_t slot = have ? (live_in ? live_in[k] : k) : 0u;

    RayRegs y;
    y.flags = 0;
    y.nf_ok = false;
    y.pt = y.pph = 0.0;
    if (have) load_ray(ws, slot, y);
    const Hole<double> bh = make_hole(P);

    bool live = have && ray_live(y);
    KsRayConsts rc;
    ray_resume<KIND, ARITH>(bh, y, P, live, rc);
    for (uint32_t it = 0; it < P.max_tries; ++it) {
        if (__ballot(live) == 0ull) break; // whole wave finished: early out
        if (live) live = advance_one<KIND, ARITH, METHOD>(bh, y, P, ws, slot, rc);
    }

    // one-launch schedule: max_tries is the hard bound no ray of a correct kernel reaches
    if (P.final_launch && live) {
        y.flags = (y.flags & ~kFlagTermMask) | GRV_TERM_MAXSTEPS;
        live = false;
    }
#if GRV_WS_RELOAD && defined(__HIP_DEVICE_COMPILE__)
    // The epilogue takes the workspace pointers from the kernel-argument segment again (`ws` is the first
    // argument: offset 0) through a pointer the optimiser cannot see through, instead of keeping a dozen
    // SGPRs live across the try loop for ten stores at the very end: 10 of the 14 v_readlane spills of
    // the FAST try loop go (+0.5-0.7 % measured A/B, profiles/r03_ab_ws_reload.jsonl).  Reloading the
    // live-list pointers the same way gains nothing more and costs the STRICT form 20 B of scratch.
    // (FAST forms only: the STRICT RKF45 forms answer the changed allocation with 20 B of scratch)
    if constexpr (ARITH == GRV_ARITH_FAST) {
        if (have) {
            const RayWorkspace *w2 = reinterpret_cast<const RayWorkspace *>(__builtin_amdgcn_kernarg_segment_ptr());
            asm volatile("" : "+s"(w2));
            store_ray(*w2, slot, y);
        }
    } else {
        if (have) store_ray(ws, slot, y);
    }
#else
    if (have) store_ray(ws, slot, y);
#endif

    // block-aggregated append of the survivors (ray compaction for the next launch):
    // per-wave ballot/popcount, one atomic per block, mbcnt-style prefix inside the wave.
    if (live_out) {
        __shared__ uint32_t s_wave_cnt[kSegBlock / 64];
        __shared__ uint32_t s_base;
        const unsigned long long mask = __ballot(live);
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        if (lane == 0) s_wave_cnt[wave] = (uint32_t)__popcll(mask);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = 0;
            for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) total += s_wave_cnt[w];
            s_base = total ? atomicAdd(live_out_count, total) : 0u;
        }
        __syncthreads();
        if (live) {
            uint32_t off = s_base;
            for (uint32_t w = 0; w < wave; ++w) off += s_wave_cnt[w];
            live_out[off + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = slot;
        }
    }
}

// ---------------------------------------------------------------------------
// The compacting schedule's launches 2..: the segment kernel's body over a live list whose LENGTH is read from
// device memory (the previous launch's append counter), so the host never waits between launches -- the whole
// frame is queued at once.  The grid is sized by the host from a forecast (the live counts the previous frame's
// launches reported, CompactFeedback) and strides over the list, so any grid is correct for any count; surplus
// blocks find nothing and exit.  Three counters rotate: launch j reads c[j % 3], appends to c[(j + 1) % 3] and
// clears c[(j + 2) % 3] (which launch j - 1 read and nobody touches during launch j) for launch j + 1 to append to.
// Same advance_one as every other schedule: results are bitwise those of the one-launch frame.
// ---------------------------------------------------------------------------
template <int KIND, int ARITH, int METHOD>
__global__ __launch_bounds__(kSegBlock) __attribute__((amdgpu_waves_per_eu(kSegmentWavesMin<KIND, METHOD>)))
void integrate_compact_kernel(
    RayWorkspace ws, SegmentParams P, const uint32_t *__restrict__ live_in, const uint32_t *__restrict__ live_in_count,
    uint32_t *__restrict__ live_out, uint32_t *__restrict__ live_out_count, uint32_t *__restrict__ clear_count,
    uint32_t *__restrict__ feedback) {
    const uint32_t n_live = *live_in_count;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *clear_count = 0u;
        if (feedback) __hip_atomic_store(feedback, n_live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // pinned host word
    }
    const Hole<double> bh = make_hole(P);
    for (uint32_t base = blockIdx.x * blockDim.x; base < n_live; base += gridDim.x * blockDim.x) {
        const uint32_t k = base + threadIdx.x;
        const bool have = k < n_live;
        const uint32_t slot = have ? (live_in ? live_in[k] : k) : 0u;
        RayRegs y;
        y.flags = 0;
        y.nf_ok = false;
        y.pt = y.pph = 0.0;
#if GRV_WS_RELOAD && defined(__HIP_DEVICE_COMPILE__)
        // FAST forms: the thirteen array pointers of the load and of the store come from the kernel-argument
        // segment again (`ws` is the first argument: offset 0) through a pointer the optimiser cannot see through,
        // right where they are needed, instead of living in SGPRs across the try loop and the stride loop (this
        // kernel parked ~60 scalars in VGPR lanes otherwise: 129 v_readlane / v_writelane per chunk)
        if constexpr (ARITH == GRV_ARITH_FAST) {
            if (have) {
                const RayWorkspace *w1 = reinterpret_cast<const RayWorkspace *>(__builtin_amdgcn_kernarg_segment_ptr());
                asm volatile("" : "+s"(w1));
                load_ray(*w1, slot, y);
            }
        } else {
            if (have) load_ray(ws, slot, y);
        }
#else
        if (have) load_ray(ws, slot, y);
#endif
        bool live = have && ray_live(y);
        KsRayConsts rc;
        ray_resume<KIND, ARITH>(bh, y, P, live, rc);
        for (uint32_t it = 0; it < P.max_tries; ++it) {
            if (__ballot(live) == 0ull) break;
            if (live) live = advance_one<KIND, ARITH, METHOD>(bh, y, P, ws, slot, rc);
        }
        if (P.final_launch && live) { // max_tries is then the hard bound no ray of a correct kernel reaches
            y.flags = (y.flags & ~kFlagTermMask) | GRV_TERM_MAXSTEPS;
            live = false;
        }
#if GRV_WS_RELOAD && defined(__HIP_DEVICE_COMPILE__)
        if constexpr (ARITH == GRV_ARITH_FAST) {
            if (have) {
                const RayWorkspace *w2 = reinterpret_cast<const RayWorkspace *>(__builtin_amdgcn_kernarg_segment_ptr());
                asm volatile("" : "+s"(w2));
                store_ray(*w2, slot, y);
            }
        } else {
            if (have) store_ray(ws, slot, y);
        }
#else
        if (have) store_ray(ws, slot, y);
#endif
        if (live_out) { // block-aggregated append of the survivors, as in the segment kernel
            __shared__ uint32_t s_wave_cnt[kSegBlock / 64];
            __shared__ uint32_t s_base;
            const unsigned long long mask = __ballot(live);
            const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
            if (lane == 0) s_wave_cnt[wave] = (uint32_t)__popcll(mask);
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t total = 0;
                for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) total += s_wave_cnt[w];
                s_base = total ? atomicAdd(live_out_count, total) : 0u;
            }
            __syncthreads();
            if (live) {
                uint32_t off = s_base;
                for (uint32_t w = 0; w < wave; ++w) off += s_wave_cnt[w];
                live_out[off + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = slot;
            }
            __syncthreads(); // s_wave_cnt / s_base are reused by the next stride
        }
    }
}

// ---------------------------------------------------------------------------
// The path kernel: integrate() with IntegrationOptions.record_path (integrator.rs:32), i.e.
// Trajectory.path (mod.rs:160).  One launch runs every ray of a (small) batch to its end and
// appends the state after every completed loop body (mod.rs:241-243: after the step, the periodic
// renormalisation and `steps += 1`) to the ray's row of `paths`, AoS GeodesicState like the
// reference's Vec<GeodesicState>.  Point 0 is the state as handed in, BEFORE the initial
// renormalize_null (mod.rs:193-197 pushes it ahead of mod.rs:200), so it comes from the caller's
// array, not from the workspace.  counts[i] = points the reference's Vec would hold (1 + steps
// taken); at most max_points of them are stored.  Same advance_one as every other schedule.
// ---------------------------------------------------------------------------
template <int KIND, int ARITH, int METHOD>
__global__ __launch_bounds__(kBlock) void integrate_path_kernel(
    RayWorkspace ws, SegmentParams P, const double *__restrict__ states_in, double *__restrict__ paths,
    uint32_t *__restrict__ counts, uint32_t max_points) {
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
    const bool have = slot < ws.n;
    RayRegs y;
    y.flags = 0;
    y.nf_ok = false;
    y.pt = y.pph = 0.0;
    if (have) load_ray(ws, slot, y);
    const Hole<double> bh = make_hole(P);
    double2 *row = reinterpret_cast<double2 *>(paths + (size_t)slot * max_points * 8);
    uint32_t npts = 0;
    if (have) {
        if (max_points > 0u) {
            const double2 *src = reinterpret_cast<const double2 *>(states_in + (size_t)slot * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) row[q] = src[q];
        }
        npts = 1;
    }
    bool live = have && ray_live(y);
    KsRayConsts rc;
    ray_resume<KIND, ARITH>(bh, y, P, live, rc);
    for (uint32_t it = 0; it < P.max_tries; ++it) {
        if (__ballot(live) == 0ull) break;
        if (live) {
            const uint32_t before = y.steps;
            live = advance_one<KIND, ARITH, METHOD>(bh, y, P, ws, slot, rc);
            if (y.steps != before) {
                if (npts < max_points) {
                    double2 *dst = row + (size_t)npts * 4;
                    dst[0] = make_double2(y.t, y.r);
                    dst[1] = make_double2(y.th, y.ph);
                    dst[2] = make_double2(y.pt, y.pr);
                    dst[3] = make_double2(y.pth, y.pph);
                }
                npts += 1;
            }
        }
    }
    if (live) y.flags = (y.flags & ~kFlagTermMask) | GRV_TERM_MAXSTEPS; // hard try bound: never a hang
    if (have) {
        store_ray(ws, slot, y);
        counts[slot] = npts;
    }
}

// ---------------------------------------------------------------------------
// The refill kernel: a resident grid (one launch) whose waves pull rays from a
// global cursor.  Every `P.max_tries` tries a wave hands its finished lanes' rays
// back to HBM and gives those lanes the next unclaimed slots (one wave-aggregated
// atomic), so an incoherent batch keeps its lanes busy without relaunching.  A ray's
// arithmetic does not depend on the lane that runs it: results are bitwise those of
// the segment kernel.
// ---------------------------------------------------------------------------
// (at least two waves per SIMD: the Boyer-Lindquist RKF45 form would otherwise take 257 registers
// and run one)
template <int KIND, int ARITH, int METHOD>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2)))
void integrate_refill_kernel(RayWorkspace ws, SegmentParams P,
                                                                  uint32_t *__restrict__ cursor) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long below = (1ull << lane) - 1ull;
    const Hole<double> bh = make_hole(P);
    RayRegs y;
    y.flags = 0;
    y.nf_ok = false;
    y.pt = y.pph = 0.0;
    KsRayConsts rc = ks_ray_consts(bh, 0.0, 0.0);
    uint32_t slot = 0;
    bool have = false, live = false, more = true; // `more` is wave-uniform
    for (;;) {
        if (more) {
            const unsigned long long need = __ballot(!live);
            if (need) {
                if (!live && have) store_ray(ws, slot, y);
                const uint32_t cnt = (uint32_t)__popcll(need);
                uint32_t base = 0;
                if (lane == (uint32_t)__ffsll((long long)need) - 1u) base = atomicAdd(cursor, cnt);
                base = __shfl(base, __ffsll((long long)need) - 1);
                more = base + cnt < ws.n;
                if (!live) {
                    const uint32_t mine = base + (uint32_t)__popcll(need & below);
                    have = mine < ws.n;
                    if (have) {
                        slot = mine;
                        load_ray(ws, slot, y);
                        live = ray_live(y);
                        ray_resume<KIND, ARITH>(bh, y, P, live, rc);
                    }
                }
            }
        }
        if (__ballot(live) == 0ull) {
            if (!more) break;
            continue; // every claimed ray was already finished: claim again
        }
        for (uint32_t it = 0; it < P.max_tries; ++it) {
            if (live) live = advance_one<KIND, ARITH, METHOD>(bh, y, P, ws, slot, rc);
            if (__ballot(live) == 0ull) break;
        }
        if (live && y.tries >= P.try_cap) { // hard bound (engine.hip try_bound): never a hang
            y.flags = (y.flags & ~kFlagTermMask) | GRV_TERM_MAXSTEPS;
            live = false;
        }
    }
    if (have) store_ray(ws, slot, y);
}

// ---------------------------------------------------------------------------
// The single-ray kernel: the whole of integrate_ray_relativistic (lib.rs:422-464) for ONE
// geodesic in one launch.  The initial state arrives in the kernel arguments (no upload), lane 0
// runs ray_begin + the try loop in registers, and the end state goes straight to a block of
// pinned host memory, followed by the call's sequence number (system-scope release) that the host
// is polling -- no workspace, no copies, no second launch.  Same advance_one as every other
// schedule, so the result is bitwise that of a 1-ray batch.
// ---------------------------------------------------------------------------
template <int KIND, int ARITH>
__global__ __launch_bounds__(64) void single_ray_kernel(SegmentParams P, SingleRayIn in, double h0,
                                                        SingleRayOut *out, uint32_t seq) {
    if (threadIdx.x != 0) return;
    RayRegs y;
    y.t = in.v[0];
    y.r = in.v[1];
    y.th = in.v[2];
    y.ph = in.v[3];
    y.pt = in.v[4];
    y.pr = in.v[5];
    y.pth = in.v[6];
    y.pph = in.v[7];
    y.h = h0;
    y.drift = 0.0;
    y.steps = 0;
    y.tries = 0;
    y.flags = kFlagValid;
    const Hole<double> bh = make_hole(P);
    ray_begin<KIND>(bh, y, P, true);
    bool live = ray_live(y);
    KsRayConsts rc;
    ray_resume<KIND, ARITH>(bh, y, P, live, rc);
    RayWorkspace ws{}; // only the crossing recorder writes to it, and P.shading == 0 here
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    while (live && y.tries < P.try_cap) live = advance_one<KIND, ARITH, GRV_METHOD_RKF45>(bh, y, P, ws, 0u, rc);
    out->loop_cycles = clock64() - c0;
    out->loop_ticks = wall_clock64() - w0;
    if (live) y.flags = (y.flags & ~kFlagTermMask) | GRV_TERM_MAXSTEPS; // hard bound reached: never a hang
    out->state[0] = y.t;
    out->state[1] = y.r;
    out->state[2] = y.th;
    out->state[3] = y.ph;
    out->state[4] = y.pt;
    out->state[5] = y.pr;
    out->state[6] = y.pth;
    out->state[7] = y.pph;
    out->drift = y.drift;
    out->steps = y.steps;
    out->tries = y.tries;
    out->term = y.flags & kFlagTermMask;
    __threadfence_system();
    __hip_atomic_store(&out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------
// init kernels
// ---------------------------------------------------------------------------
// batch: AoS GeodesicState [n][8] -> SoA workspace (geodesic/mod.rs:23-30 layout)
template <int KIND>
__global__ __launch_bounds__(kBlock) void init_from_states_kernel(
    RayWorkspace ws, SegmentParams P, const double *__restrict__ states, double h0, int adaptive) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= ws.n) return;
    const double2 *src = reinterpret_cast<const double2 *>(states + (size_t)i * 8);
    const double2 a = src[0], b = src[1], c = src[2], d = src[3];
    RayRegs y;
    y.t = a.x;
    y.r = a.y;
    y.th = b.x;
    y.ph = b.y;
    y.pt = c.x;
    y.pr = c.y;
    y.pth = d.x;
    y.pph = d.y;
    y.h = h0;
    y.drift = 0.0;
    y.steps = 0;
    y.tries = 0;
    y.flags = kFlagValid;
    const Hole<double> bh = make_hole(P);
    ray_begin<KIND>(bh, y, P, adaptive != 0);
    store_ray(ws, i, y);
    ws.pt[i] = y.pt;
    ws.pph[i] = y.pph;
}

// slot -> pixel of this rank's tile set.  A slot block of 4096 is one 64x64 tile;
// inside it every 64 consecutive slots (one wave) cover an 8x8 pixel block, the
// launch shape of the reference's compute pass (compute.wgsl.ts:147).
__device__ __forceinline__ bool slot_to_pixel(const FrameGeom &G, uint32_t slot, uint32_t &X,
                                              uint32_t &Y, uint32_t &out_index) {
    const uint32_t tile_local = slot >> 12;
    const uint32_t within = slot & 4095u;
    const uint32_t w = within >> 6, lane = within & 63u;
    const uint32_t px = ((w & 7u) << 3) + (lane & 7u);
    const uint32_t py = ((w >> 3) << 3) + (lane >> 3);
    const uint32_t tile = tile_local * G.tile_world + G.tile_rank;
    const uint32_t tx = tile % G.tiles_x, ty = tile / G.tiles_x;
    X = tx * 64u + px;
    Y = ty * 64u + py;
    out_index = (G.tile_world <= 1u) ? (Y * G.width + X) : (tile_local * 4096u + py * 64u + px);
    return tile_local < G.n_tiles_local && ty < G.tiles_y && X < G.width && Y < G.height;
}

__device__ __forceinline__ void mat4_mul(const double *m, double x, double y, double z, double w,
                                         double out[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = m[0 + r] * x + m[4 + r] * y + m[8 + r] * z + m[12 + r] * w;
}

// pixel -> Kerr-Schild initial state: src/shaders/compute.wgsl.ts:159-187 in f64
template <int KIND>
__global__ __launch_bounds__(kBlock) void init_from_pixels_kernel(RayWorkspace ws, SegmentParams P,
                                                                  FrameGeom G, CameraDev cam,
                                                                  double h0, int adaptive) {
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
    if (slot >= ws.n) return;
    uint32_t X, Y, oi;
    const bool valid = slot_to_pixel(G, slot, X, Y, oi);
    RayRegs y;
    y.drift = 0.0;
    y.steps = 0;
    y.tries = 0;
    y.h = h0;
    if (!valid) {
        y.t = y.r = y.th = y.ph = y.pr = y.pth = y.pt = y.pph = 0.0;
        y.flags = 0;
        store_ray(ws, slot, y);
        ws.pt[slot] = 0.0;
        ws.pph[slot] = 0.0;
        return;
    }
    const double ux = ((double)X + cam.off[0]) / (double)G.width;
    const double uy = ((double)Y + cam.off[1]) / (double)G.height;
    const double ndc_x = ux * 2.0 - 1.0;
    const double ndc_y = uy * 2.0 - 1.0;
    double vt[4];
    mat4_mul(cam.inv_proj, ndc_x, -ndc_y, 1.0, 1.0, vt);
    double vx = vt[0] / vt[3], vy = vt[1] / vt[3], vz = vt[2] / vt[3];
    double len = sqrt(vx * vx + vy * vy + vz * vz);
    vx /= len;
    vy /= len;
    vz /= len;
    double wd[4];
    mat4_mul(cam.inv_view, vx, vy, vz, 0.0, wd);
    len = sqrt(wd[0] * wd[0] + wd[1] * wd[1] + wd[2] * wd[2]);
    const double wx = wd[0] / len, wy = wd[1] / len, wz = wd[2] / len;

    // the camera's own (r, theta, phi) and their sines/cosines are per-frame constants: formed
    // once on the host (engine.hip: camera_constants) instead of 8.3 M times here
    const double r0 = cam.r0, theta0 = cam.theta0, phi0 = cam.phi0;
    const double st = cam.st, ct = cam.ct, sp = cam.sp, cp = cam.cp;
    const double pr_far = wx * (st * cp) + wy * ct + wz * (st * sp);
    const double pth_far = (wx * (ct * cp) + wy * (-st) + wz * (ct * sp)) / r0;
    const double safe_st = fmax(st, 1e-4);
    const double pph_far = (wx * (-sp) + wy * 0.0 + wz * cp) / (r0 * safe_st);

    y.t = 0.0;
    y.r = r0;
    y.th = theta0;
    y.ph = phi0;
    y.pt = -1.0;
    y.pr = pr_far;
    y.pth = pth_far * r0 * r0;
    y.pph = pph_far * r0 * r0 * st * st;
    y.flags = kFlagValid;
    const Hole<double> bh = make_hole(P);
    ray_begin<KIND>(bh, y, P, adaptive != 0);
    store_ray(ws, slot, y);
    ws.pt[slot] = y.pt;
    ws.pph[slot] = y.pph;
}

} // namespace
