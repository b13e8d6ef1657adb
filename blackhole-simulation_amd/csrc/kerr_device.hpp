// kerr_device.hpp -- device-side Kerr / Schwarzschild Hamiltonian right-hand side
// for gfx950.  Everything has internal linkage: this header is compiled twice,
// once with -ffp-contract=off (STRICT translation unit) and once with
// -ffp-contract=fast (FAST translation unit).
//
// Reference behaviour restated (paths under physics-engine/gravitas-core/src):
//   get_state_derivative        geodesic/hamiltonian.rs:13-35
//   Kerr::contravariant_ks      metric/kerr.rs:412-440
//   Kerr::hamiltonian_derivs_ks metric/kerr.rs:442-499
//   Kerr::contravariant_bl      metric/kerr.rs:266-293
//   Kerr::hamiltonian_derivs_bl metric/kerr.rs:295-372
//   Schwarzschild               metric/schwarzschild.rs:62-111
//   hamiltonian                 invariants/mod.rs:25-37
//   renormalize_null            invariants/renormalization.rs:13-45
//
// Only the six evolving components (t, r, theta, phi, p_r, p_theta) are carried:
// dp_t/dl = dp_phi/dl = 0 exactly (hamiltonian.rs:33), so p_t and p_phi are
// per-ray constants and every `p += 0 * s` of the reference is the identity.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/gravitas_abi.h"
#if defined(GRV_SPECIFIED_LIBM)
#include "strict_libm.hpp"
#endif

namespace {

template <typename T> struct Hole {
    T M;  // mass
    T a;  // spin * mass
    T a2; // a * a
    T two_m; // 2 M
    bool divs_ok = false; // M and a admit the shared-reciprocal division form (divs_ok_hole)
    bool divs_nf = false; // ... and its form without v_div_fixup (divs_nf_hole; per ray: RayRegs::nf_ok)
};

template <typename T> struct Deriv {
    T dt, dr, dth, dph, dpr, dpth;
};

// metric components shared by the Hamiltonian and the null renormalisation
template <typename T> struct GInv {
    T tt, tr, tph, rr, thth, phph, rph;
};

template <typename T> __device__ __forceinline__ T fmax_t(T a, T b);
template <> __device__ __forceinline__ double fmax_t<double>(double a, double b) { return fmax(a, b); }
template <> __device__ __forceinline__ float fmax_t<float>(float a, float b) { return fmaxf(a, b); }
template <typename T> __device__ __forceinline__ T fabs_t(T a);
template <> __device__ __forceinline__ double fabs_t<double>(double a) { return fabs(a); }
template <> __device__ __forceinline__ float fabs_t<float>(float a) { return fabsf(a); }
template <typename T> __device__ __forceinline__ T sqrt_t(T a);
template <> __device__ __forceinline__ double sqrt_t<double>(double a) { return sqrt(a); }
template <> __device__ __forceinline__ float sqrt_t<float>(float a) { return sqrtf(a); }
template <typename T> __device__ __forceinline__ void sincos_t(T x, T *s, T *c);
// f64 transcendental functions of the ray path.  The STRICT unit (kernels_strict.hip) defines
// GRV_SPECIFIED_LIBM and gets the written-out routines of strict_libm.hpp, whose results are a
// pure function of the argument (the checker in oracle/ evaluates the same published algorithms);
// the FAST unit's reference-order metrics (BL, Schwarzschild) use the device library.
#if defined(GRV_SPECIFIED_LIBM)
template <> __device__ __forceinline__ void sincos_t<double>(double x, double *s, double *c) { strictm::sl_sincos(x, s, c); }
__device__ __forceinline__ double pow_rs(double x, double y) { return strictm::sl_pow(x, y); }
__device__ __forceinline__ double exp_rs(double x) { return strictm::sl_exp(x); }
#else
template <> __device__ __forceinline__ void sincos_t<double>(double x, double *s, double *c) { sincos(x, s, c); }
__device__ __forceinline__ double pow_rs(double x, double y) { return pow(x, y); }
__device__ __forceinline__ double exp_rs(double x) { return exp(x); }
#endif
template <> __device__ __forceinline__ void sincos_t<float>(float x, float *s, float *c) { sincosf(x, s, c); }

// ---------------------------------------------------------------------------
// FAST-contract math: short, branch-free replacements for the OCML calls on the
// hot path.  They are only instantiated in the -ffp-contract=fast unit.
// ---------------------------------------------------------------------------

// sin/cos for |x| << 2^20 * pi/2: three-constant Cody-Waite reduction by pi/2 with
// fma, then the fdlibm __kernel_sin / __kernel_cos minimax polynomials on
// [-pi/4, pi/4].  <= 1 ulp-level error; no large-argument (Payne-Hanek) path --
// theta of a geodesic is O(1..100).
__device__ __forceinline__ void fast_sincos(double x, double *sn, double *cs) {
    const double j = rint(x * 6.36619772367581382433e-01); // 2/pi
    double r = fma(-j, 1.57079632679489655800e+00, x);
    r = fma(-j, 6.12323399573676603587e-17, r);
    r = fma(-j, -1.49738490485916983765e-33, r); // pi/2 = hi + mid + lo (lo < 0)
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double sr = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double cr = fma(z, fma(z, pc, -0.5), 1.0);
    const int q = (int)j;
    const double s0 = (q & 1) ? cr : sr;
    const double c0 = (q & 1) ? sr : cr;
    *sn = (q & 2) ? -s0 : s0;
    *cs = ((q + 1) & 2) ? -c0 : c0;
}

// A/B switches of the STRICT division forms (TU=strict tools/ab_build.sh builds the other forms)
#ifndef GRV_STRICT_NOFIXUP
#define GRV_STRICT_NOFIXUP 1 // SharedDivNoFixup ahead of SharedDiv
#endif
#ifndef GRV_STRICT_CONSTDEN
#define GRV_STRICT_CONSTDEN 1 // ConstDen for the Fehlberg stage scales
#endif
// A/B switches of the FAST Kerr-Schild geometry (tools/ab_build.sh builds the other forms)
#ifndef GRV_TRIG_BITS
#define GRV_TRIG_BITS 1
#endif
#ifndef GRV_POLAR_HI_WORD
#define GRV_POLAR_HI_WORD 1
#endif

// Bit-level helpers for the quadrant logic of the FAST trigonometry.  Written as single
// instructions on purpose: the optimiser turns the portable forms ((a & m) | (b & ~m) with a
// sign-extended bit m, x + (y << 31)) back into compare + select pairs, which is exactly the
// instruction count these exist to avoid.
__device__ __forceinline__ int bits_sext_bit0(int x) { return __builtin_amdgcn_sbfe(x, 0, 1); } // v_bfe_i32
__device__ __forceinline__ int bits_select(int mask, int a, int b) { // (a & mask) | (b & ~mask): v_bfi_b32
    int d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(mask), "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ int bits_shl_add(int x, int sh, int y) { // (x << sh) + y: v_lshl_add_u32
    int d;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "n"(sh), "v"(y));
    return d;
}

// 1/x: v_rcp_f64 seed (~2^-23 relative) + two Newton steps -> ~1 ulp.
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    return y;
}
// same seed, one third-order step y (1 + e + e^2): residual e^3 ~ 2^-69, one fma less
__device__ __forceinline__ double fast_rcp3(double x) {
    const double y = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, y, 1.0);
    return fma(y, fma(e, e, e), y);
}

// x^(-1/5) and x^(-1/4) for the step-size controller: f32 exp2/log2 seed (~2e-7) + one
// Newton step on y^-n = x (error -> ~3 e^2 ~ 1e-13).  The result only scales the next
// step size: a 1e-13 relative change of h moves the next sample point along the same
// trajectory by 1e-13 h and leaves the local error (hence the trajectory) unchanged to
// far below the integrator's tolerance.
// x = +inf or huge -> 0 (or NaN), which the caller's fmax(.., 0.1) maps to 0.1.
__device__ __forceinline__ double fast_pow_m1_5(double x) {
    const double y = (double)__builtin_amdgcn_exp2f(-0.2f * __builtin_amdgcn_logf((float)x));
    const double y2 = y * y;
    const double y5 = y2 * y2 * y;
    return y * fma(-0.2 * x, y5, 1.2);
}
__device__ __forceinline__ double fast_pow_m1_4(double x) {
    const double y = (double)__builtin_amdgcn_exp2f(-0.25f * __builtin_amdgcn_logf((float)x));
    const double y2 = y * y;
    const double y4 = y2 * y2;
    return y * fma(-0.25 * x, y4, 1.25);
}

// ---------------------------------------------------------------------------
// Division policies of the reference-order code.
//
// IeeeDiv: the language's `/` -- for f64 the compiler's correctly rounded expansion
//   (v_div_scale x2, v_rcp, five fma, v_div_fmas, v_div_fixup: ten instructions a quotient).
// SharedDiv (f64, STRICT Kerr-Schild): the SAME expansion with the part that depends on the
//   denominator alone -- the reciprocal seed and its two Newton steps -- evaluated once per
//   denominator and shared by every quotient over it (the Kerr-Schild right-hand side has fifteen
//   quotients over five denominators).  Per quotient that leaves the product, the residual fma, the
//   correction fma and v_div_fixup, which maps zero / infinite / NaN operands exactly as the full
//   sequence does.  The two v_div_scale steps are dropped: they are the identity (and v_div_fmas a
//   plain fma) whenever both operands are zero, non-finite or moderate in magnitude, and `divs_ok`
//   below admits a right-hand side to this form only then.  Inside that range the instruction
//   sequence per quotient is the compiler's own, so the quotient has the same bits; outside it the
//   wave takes the IeeeDiv form.  Which form ran is therefore invisible in the results.
// SharedDivNoFixup (round 3): SharedDiv without the closing v_div_fixup either.  For finite non-zero
//   operands of moderate magnitude v_div_fixup returns |q| with the sign of n/d, i.e. q itself; it only
//   matters for zero, infinite or NaN operands.  `divs_nf` admits a right-hand side to this form when
//   M, r are in [2^-20, 2^20] (positive), |a| in [2^-20, 2^20] and |sin|, |cos| >= 2^-70: then every
//   operand is finite, every denominator positive, and the only zero a numerator can be is the +0 of an
//   exact cancellation (delta, sigma - 2 r^2, the two-term numerators), for which the sequence
//   gives q0 = +0, res = +0, q = +0 = (+0)/d.  A numerator of -0 would come out as +0, so the two
//   numerators the reference negates AFTER a possible cancellation are divided before the negation
//   (-(x)/d == -(x/d) bit for bit) and the one product whose zero can take either sign
//   (delta * dSigma/dtheta) keeps the fixup (`div_signed`).  rhs_ref_at's NOFIX branches hold the
//   case analysis next to each quotient.  Checked alone by GRV_MATH_DIV_NOFIX (tests/test_ref_libm.py).
// kExactFma: the guarded forms may also contract a product that is exact by construction in their
//   operand range (2 * x) into the neighbouring sum: fma(2, w, u) == u + 2 w when 2 w is exact.
// ---------------------------------------------------------------------------
struct IeeeDiv {
    static constexpr bool kNoFixup = false;
    template <typename T> struct Den { T d; };
    template <typename T> static __device__ __forceinline__ Den<T> prep(T d) { return {d}; }
    template <typename T> static __device__ __forceinline__ T div(T n, const Den<T> &D) { return n / D.d; }
    template <typename T> static __device__ __forceinline__ T div_signed(T n, const Den<T> &D) { return n / D.d; }
};
struct SharedDiv {
    static constexpr bool kNoFixup = false;
    template <typename T> struct Den { T d, r; };
    static __device__ __forceinline__ Den<double> prep(double d) {
        // r2 of the f64 fdiv expansion: rcp seed + two Newton steps
        const double r0 = __builtin_amdgcn_rcp(d);
        const double e0 = fma(-d, r0, 1.0);
        const double r1 = fma(r0, e0, r0);
        const double e1 = fma(-d, r1, 1.0);
        return {d, fma(r1, e1, r1)};
    }
    static __device__ __forceinline__ double div(double n, const Den<double> &D) {
        const double q0 = n * D.r;
        const double res = fma(-D.d, q0, n);
        return __builtin_amdgcn_div_fixup(fma(res, D.r, q0), D.d, n);
    }
    static __device__ __forceinline__ double div_signed(double n, const Den<double> &D) { return div(n, D); }
};
struct SharedDivNoFixup {
    static constexpr bool kNoFixup = true;
    template <typename T> using Den = SharedDiv::Den<T>;
    static __device__ __forceinline__ Den<double> prep(double d) { return SharedDiv::prep(d); }
    // n finite and +0 or non-zero, d > 0, both moderate
    static __device__ __forceinline__ double div(double n, const Den<double> &D) {
        const double q0 = n * D.r;
        const double res = fma(-D.d, q0, n);
        return fma(res, D.r, q0);
    }
    // n may be -0
    static __device__ __forceinline__ double div_signed(double n, const Den<double> &D) { return SharedDiv::div(n, D); }
};

// n / D for a compile-time D that is not a power of two (the Fehlberg tableau's 2197, 216, 513, 4104,
// 27, 2565, 40): SharedDiv with the denominator's refined reciprocal as a literal.  kR2 must be the value
// SharedDiv::prep(D).r takes on the device -- it is 1/D correctly rounded for each of the seven, which
// GRV_MATH_RCP_R2 lets tests/test_ref_libm.py assert on the GPU -- so the instruction sequence per
// quotient is again the compiler's own minus the two v_div_scale steps (the identity for a numerator
// that is zero, non-finite or in [2^-100, 2^113]: const_div_ok).  v_div_fixup stays: the step size may
// be a signed zero or NaN.
template <int D> struct ConstDen {
    static constexpr double kD = (double)D;
    static constexpr double kR2 = 1.0 / (double)D;
    static __device__ __forceinline__ double div(double n) {
        const double q0 = n * kR2;
        const double res = fma(-kD, q0, n);
        return __builtin_amdgcn_div_fixup(fma(res, kR2, q0), kD, n);
    }
};
// the step size h admits ConstDen for c * h (|c| <= 7296): zero, infinite, NaN or 2^-100 <= |h| <= 2^100
__device__ __forceinline__ bool const_div_ok(double h) {
    constexpr int kZeroInfNan = 0x001 | 0x002 | 0x004 | 0x200 | 0x020 | 0x040;
    return __builtin_amdgcn_class(h, kZeroInfNan) || (fabs(h) >= 0x1p-100 && fabs(h) <= 0x1p100);
}

// x is zero or lo <= |x| <= hi  (false for NaN / infinities)
__device__ __forceinline__ bool zero_or_within(double x, double lo, double hi) {
    const double ax = fabs(x);
    return ax == 0.0 || (ax >= lo && ax <= hi);
}
// Operand range in which every quotient of the Kerr-Schild metric and its derivatives keeps both
// v_div_scale steps the identity: M, a, r zero or in [2^-20, 2^20]; sin, cos zero or >= 2^-70
// (cos(pi/2) = 6e-17 = 2^-54 is the common equatorial value).  Then every denominator is zero or
// in [2^-266, 2^82], every numerator zero or in [2^-290, 2^83] (differences of products are zero
// or at least an ulp of their larger term): all normal, numerator exponents above the 2^-969 line
// of v_div_scale, exponent differences far inside +-768.
__device__ __forceinline__ bool divs_ok_hole(double M, double a) {
    return zero_or_within(M, 0x1p-20, 0x1p20) && zero_or_within(a, 0x1p-20, 0x1p20);
}
__device__ __forceinline__ bool divs_ok_point(double r, double sin_theta, double cos_theta) {
    return zero_or_within(r, 0x1p-20, 0x1p20) && zero_or_within(sin_theta, 0x1p-70, 1.0) &&
           zero_or_within(cos_theta, 0x1p-70, 1.0);
}

// The stricter range of SharedDivNoFixup: no zeros, positive M and r (so that every denominator is
// positive and no numerator is a signed zero other than the +0 of a cancellation).  Four compares
// per point instead of the nine of divs_ok_point; NaN fails every one of them.
__device__ __forceinline__ bool divs_nf_hole(double M, double a) {
    return M >= 0x1p-20 && M <= 0x1p20 && fabs(a) >= 0x1p-20 && fabs(a) <= 0x1p20;
}
// per-ray part of the admission: p_t and p_phi (constants of the motion) non-zero and moderate.  Then
// g^tt p_t (g^tt <= -1) and g^phph p_phi are non-zero normal numbers, which is what lets the NOFIX
// forms drop the products with the structurally zero g^{t phi} (see rhs_ref_at)
__device__ __forceinline__ bool divs_nf_consts(double p_t, double p_ph) {
    return fabs(p_t) >= 0x1p-500 && fabs(p_t) <= 0x1p500 && fabs(p_ph) >= 0x1p-500 && fabs(p_ph) <= 0x1p500;
}
__device__ __forceinline__ bool divs_nf_point(double r, double sin_theta, double cos_theta) {
    return r >= 0x1p-20 && r <= 0x1p20 && fabs(sin_theta) >= 0x1p-70 && fabs(cos_theta) >= 0x1p-70;
}

// ---------------------------------------------------------------------------
// Reference-order inverse metric.  KIND = GRV_METRIC_*.
// ---------------------------------------------------------------------------
template <int KIND, typename T, typename DIV = IeeeDiv>
__device__ __forceinline__ GInv<T> contravariant_ref(const Hole<T> &bh, T r, T sin_theta,
                                                     T cos_theta) {
    GInv<T> g;
    g.tr = T(0);
    g.tph = T(0);
    g.rph = T(0);
    const T m = bh.M;
    if constexpr (KIND == GRV_METRIC_KERR_KS) {
        const T r2 = r * r;
        const T sin2 = fmax_t(sin_theta * sin_theta, T(1e-12));
        const T cos2 = T(1) - sin2;
        const T sigma = r2 + bh.a2 * cos2;
        const T delta = r2 - T(2) * m * r + bh.a2;
        const auto by_sigma = DIV::prep(sigma);
        const auto by_sigma_sin2 = DIV::prep(sigma * sin2);
        g.tt = -(T(1) + DIV::div(T(2) * m * r, by_sigma));
        g.tr = DIV::div(T(2) * m * r, by_sigma);
        g.rr = DIV::div(delta, by_sigma);
        g.thth = DIV::div(T(1), by_sigma);
        g.phph = DIV::div(T(1), by_sigma_sin2);
        g.rph = DIV::div(bh.a, by_sigma);
    } else if constexpr (KIND == GRV_METRIC_KERR_BL) {
        const T r2 = r * r;
        const T sin2 = sin_theta * sin_theta;
        const T cos2 = cos_theta * cos_theta;
        const T sigma = r2 + bh.a2 * cos2;
        const T delta = r2 - T(2) * m * r + bh.a2;
        g.tt = -((sigma * (r2 + bh.a2) + T(2) * m * r * bh.a2 * sin2) / (delta * sigma));
        g.rr = delta / sigma;
        g.thth = T(1) / sigma;
        g.phph = (sin2 < T(1e-9)) ? T(0) : (delta - bh.a2 * sin2) / (delta * sigma * sin2);
        g.tph = -(T(2) * m * r * bh.a) / (delta * sigma);
    } else {
        const T rs = T(2) * m;
        const T sin2 = fmax_t(sin_theta * sin_theta, T(1e-12));
        g.tt = T(-1) / (T(1) - rs / r);
        g.rr = T(1) - rs / r;
        g.thth = T(1) / (r * r);
        g.phph = T(1) / (r * r * sin2);
    }
    return g;
}

// ---------------------------------------------------------------------------
// Reference-order right-hand side (used by the STRICT build for every metric
// and by the FAST build for BL / Schwarzschild).
// ---------------------------------------------------------------------------
// (the point's sine, cosine and inverse metric come from the caller: the post-step bookkeeping
// has them already when it forms stage 1 of the next try)
template <int KIND, typename T, typename DIV = IeeeDiv>
__device__ __forceinline__ Deriv<T> rhs_ref_at(const Hole<T> &bh, T r, T sin_theta, T cos_theta,
                                               const GInv<T> &g, T p_t, T p_r, T p_th, T p_ph) {
    Deriv<T> d;
    const T m = bh.M;
    const T a = bh.a;
    const T a2 = bh.a2;

    if constexpr (KIND == GRV_METRIC_KERR_KS) {
        // the contraction of get_state_derivative keeps the structurally zero entries of the
        // metric (hamiltonian.rs:19-28): 0 * p is not 0 for a non-finite momentum, and x + 0 loses
        // the sign of x = -0
        if constexpr (DIV::kNoFixup) {
            // ... unless neither can happen (divs_nf_consts: p_t, p_phi non-zero and moderate).
            // dt: x = g^tt p_t + g^tr p_r has a non-zero first term, so it is non-zero or the +0 of a
            // cancellation, never -0, and x + (+-0) == x.  dphi: (+-0 + y) + w with w = g^phph p_phi
            // non-zero: +-0 + y == y for y != 0, and for y == +-0 the sum is a zero of some sign that
            // w absorbs.  Same bits, four instructions fewer
            d.dt = g.tt * p_t + g.tr * p_r;
            d.dph = g.rph * p_r + g.phph * p_ph;
        } else {
            d.dt = g.tt * p_t + g.tr * p_r + g.tph * p_ph;
            d.dph = g.tph * p_t + g.rph * p_r + g.phph * p_ph;
        }
        d.dr = g.tr * p_t + g.rr * p_r + g.rph * p_ph;
        d.dth = g.thth * p_th;

        const T r2 = r * r;
        const T sin2 = fmax_t(sin_theta * sin_theta, T(1e-12));
        const T cos2 = T(1) - sin2;
        const T sigma = r2 + a2 * cos2;
        const T sigma2 = sigma * sigma;
        const T delta = r2 - T(2) * m * r + a2;

        const T dsigma_dr = T(2) * r;
        const T dsigma_dtheta = T(-2) * a2 * sin_theta * cos_theta;
        const T ddelta_dr = T(2) * r - T(2) * m;

        const auto by_sigma2 = DIV::prep(sigma2);
        const auto by_sigma2_sin2 = DIV::prep(sigma2 * sin2);
        const auto by_sigma2_sin4 = DIV::prep(sigma2 * sin2 * sin2);
        T dg_tt_dr, dg_tr_dr, dg_phph_dtheta;
        if constexpr (DIV::kNoFixup) {
            // (SharedDivNoFixup; operand range of divs_nf: M, r > 0, a, sin, cos != 0, all moderate.)
            // sigma - r (2r): r (2r) = 2 r^2 exactly in this range, so fma(-2, r^2, sigma) is the
            // reference's difference; it is +0 or non-zero (sigma > 0), and so is 2M times it.  The
            // reference negates BEFORE dividing, which would make a cancelled numerator -0: divide
            // first (-(x)/d == -(x/d) bit for bit)
            dg_tr_dr = DIV::div(T(2) * m * fma(T(-2), r2, sigma), by_sigma2);
            dg_tt_dr = -dg_tr_dr;
            // dsigma_dtheta sin^2 + ((2 sigma) sin) cos: the second term is 2 ((sigma sin) cos)
            // exactly (no underflow: >= 2^-180), so fma(2, w, u) is the reference's sum; a sum of two
            // non-zero terms is +0 or non-zero; negated after the division as above
            dg_phph_dtheta = -DIV::div(fma(T(2), sigma * sin_theta * cos_theta, dsigma_dtheta * sin2), by_sigma2_sin4);
        } else {
            dg_tt_dr = DIV::div(-(T(2) * m * (sigma - r * dsigma_dr)), by_sigma2);
            dg_tr_dr = -dg_tt_dr;
            dg_phph_dtheta =
                DIV::div(-(dsigma_dtheta * sin2 + sigma * T(2) * sin_theta * cos_theta), by_sigma2_sin4);
        }
        // 2 M r dsigma_dtheta, 2r, dsigma_dtheta, a 2r, a dsigma_dtheta: products of non-zero
        // moderate factors under divs_nf -- never zero.  ddelta_dr sigma - delta 2r: a difference
        // whose minuend cannot be -0 (2r - 2M cancels to +0): +0 or non-zero.
        const T dg_tt_dtheta = DIV::div(T(2) * m * r * dsigma_dtheta, by_sigma2);
        const T dg_tr_dtheta = -dg_tt_dtheta;
        const T dg_rr_dr = DIV::div(ddelta_dr * sigma - delta * dsigma_dr, by_sigma2);
        // delta = +0 on a horizon radius and dsigma_dtheta of either sign: the one numerator that can be
        // -0 either side of the negation -- keeps the fixup
        const T dg_rr_dtheta = DIV::div_signed(-(delta * dsigma_dtheta), by_sigma2);
        const T dg_thth_dr = DIV::div(-dsigma_dr, by_sigma2);
        const T dg_thth_dtheta = DIV::div(-dsigma_dtheta, by_sigma2);
        const T dg_phph_dr = DIV::div(-dsigma_dr, by_sigma2_sin2);
        const T dg_rph_dr = DIV::div(-(a * dsigma_dr), by_sigma2);
        const T dg_rph_dtheta = DIV::div(-(a * dsigma_dtheta), by_sigma2);

        const T dh_dr = T(0.5) * (dg_tt_dr * p_t * p_t + dg_rr_dr * p_r * p_r +
                                  dg_thth_dr * p_th * p_th + dg_phph_dr * p_ph * p_ph +
                                  T(2) * dg_tr_dr * p_t * p_r + T(2) * dg_rph_dr * p_r * p_ph);
        T dh_dtheta = T(0.5) * (dg_tt_dtheta * p_t * p_t + dg_rr_dtheta * p_r * p_r +
                                dg_thth_dtheta * p_th * p_th + dg_phph_dtheta * p_ph * p_ph +
                                T(2) * dg_tr_dtheta * p_t * p_r + T(2) * dg_rph_dtheta * p_r * p_ph);
        if (fabs_t(sin_theta) < T(1e-10)) dh_dtheta = T(0);
        d.dpr = -dh_dr;
        d.dpth = -dh_dtheta;
    } else if constexpr (KIND == GRV_METRIC_KERR_BL) {
        // the contraction of get_state_derivative keeps the structurally zero entries of the
        // metric (hamiltonian.rs:19-28): 0 * p is not 0 for a non-finite momentum, and x + 0 loses
        // the sign of x = -0
        d.dt = g.tt * p_t + g.tr * p_r + g.tph * p_ph;
        d.dr = g.tr * p_t + g.rr * p_r + g.rph * p_ph;
        d.dth = g.thth * p_th;
        d.dph = g.tph * p_t + g.rph * p_r + g.phph * p_ph;

        const T r2 = r * r;
        const T sin2 = sin_theta * sin_theta;
        const T cos2 = cos_theta * cos_theta;
        const T sigma = r2 + a2 * cos2;
        const T delta = r2 - T(2) * m * r + a2;
        const T sigma_sq = sigma * sigma;

        const T dsigma_dr = T(2) * r;
        const T dsigma_dtheta = T(-2) * a2 * cos_theta * sin_theta;
        const T ddelta_dr = T(2) * r - T(2) * m;

        const T dg_rr_dr = (ddelta_dr * sigma - delta * dsigma_dr) / sigma_sq;
        const T dg_rr_dtheta = -(delta * dsigma_dtheta) / sigma_sq;
        const T dg_thth_dr = -dsigma_dr / sigma_sq;
        const T dg_thth_dtheta = -dsigma_dtheta / sigma_sq;

        const T num_tphi = T(-2) * m * r * a;
        const T den_tphi = delta * sigma;
        const T dnum_tphi_dr = T(-2) * m * a;
        const T dden_tphi_dr = ddelta_dr * sigma + delta * dsigma_dr;
        const T dg_tphi_dr =
            (dnum_tphi_dr * den_tphi - num_tphi * dden_tphi_dr) / (den_tphi * den_tphi);
        const T dden_tphi_dtheta = delta * dsigma_dtheta;
        const T dg_tphi_dtheta = -(num_tphi * dden_tphi_dtheta) / (den_tphi * den_tphi);

        const T du_dr = dsigma_dr * (r2 + a2) + sigma * T(2) * r + T(2) * m * a2 * sin2;
        const T dv_dr = dden_tphi_dr;
        const T u_val = sigma * (r2 + a2) + T(2) * m * r * a2 * sin2;
        const T dg_tt_dr = -(du_dr * den_tphi - u_val * dv_dr) / (den_tphi * den_tphi);

        const T du_dtheta =
            dsigma_dtheta * (r2 + a2) + T(2) * m * r * a2 * T(2) * sin_theta * cos_theta;
        const T dv_dtheta = dden_tphi_dtheta;
        const T dg_tt_dtheta = -(du_dtheta * den_tphi - u_val * dv_dtheta) / (den_tphi * den_tphi);

        const T da_dr = -dsigma_dr / (sigma_sq * sin2);
        const T db_dr = -a2 * dden_tphi_dr / (den_tphi * den_tphi);
        const T dg_phph_dr = da_dr - db_dr;

        const T d_denom_a_dtheta = dsigma_dtheta * sin2 + sigma * T(2) * sin_theta * cos_theta;
        const T da_dtheta = -d_denom_a_dtheta / (sigma_sq * sin2 * sin2);
        const T db_dtheta = -a2 * dden_tphi_dtheta / (den_tphi * den_tphi);
        const T dg_phph_dtheta = da_dtheta - db_dtheta;

        const T dh_dr =
            T(0.5) * (p_t * p_t * dg_tt_dr + p_r * p_r * dg_rr_dr + p_th * p_th * dg_thth_dr +
                      p_ph * p_ph * dg_phph_dr + T(2) * p_t * p_ph * dg_tphi_dr);
        const T dh_dtheta = T(0.5) * (p_t * p_t * dg_tt_dtheta + p_r * p_r * dg_rr_dtheta +
                                      p_th * p_th * dg_thth_dtheta + p_ph * p_ph * dg_phph_dtheta +
                                      T(2) * p_t * p_ph * dg_tphi_dtheta);
        d.dpr = -dh_dr;
        d.dpth = -dh_dtheta;
    } else {
        // the contraction of get_state_derivative keeps the structurally zero entries of the
        // metric (hamiltonian.rs:19-28): 0 * p is not 0 for a non-finite momentum, and x + 0 loses
        // the sign of x = -0
        d.dt = g.tt * p_t + g.tr * p_r + g.tph * p_ph;
        d.dr = g.tr * p_t + g.rr * p_r + g.rph * p_ph;
        d.dth = g.thth * p_th;
        d.dph = g.tph * p_t + g.rph * p_r + g.phph * p_ph;

        const T r2 = r * r;
        const T r3 = r2 * r;
        const T sin2 = sin_theta * sin_theta;
        const T f = T(1) - T(2) * m / r;
        const T dg_tt_dr = T(-2) * m / (r2 * f * f);
        const T dg_rr_dr = T(2) * m / r2;
        const T dg_thth_dr = T(-2) / r3;
        const T dg_phph_dr = (sin2 < T(1e-12)) ? T(0) : T(-2) / (r3 * sin2);
        const T dg_phph_dtheta =
            (sin2 < T(1e-12)) ? T(0) : T(-2) * cos_theta / (r2 * sin_theta * sin2);
        const T dh_dr = T(0.5) * (dg_tt_dr * p_t * p_t + dg_rr_dr * p_r * p_r +
                                  dg_thth_dr * p_th * p_th + dg_phph_dr * p_ph * p_ph);
        const T dh_dtheta = T(0.5) * dg_phph_dtheta * p_ph * p_ph;
        d.dpr = -dh_dr;
        d.dpth = -dh_dtheta;
    }
    return d;
}

// nf_ok: this ray's constants admit the NOFIX forms (bh.divs_nf && divs_nf_consts, formed once per ray)
template <int KIND, typename T>
__device__ __forceinline__ Deriv<T> rhs_ref(const Hole<T> &bh, T r, T theta, T p_t, T p_r,
                                            T p_th, T p_ph, bool nf_ok = false) {
    T sin_theta, cos_theta;
    sincos_t(theta, &sin_theta, &cos_theta);
    if constexpr (KIND == GRV_METRIC_KERR_KS && sizeof(T) == 8) {
        // wave-uniform choice of the division form (see SharedDiv): same bits either way
        const bool nf = GRV_STRICT_NOFIXUP && nf_ok && divs_nf_point(r, sin_theta, cos_theta);
        if (GRV_STRICT_NOFIXUP && __ballot(!nf) == 0ull) {
            const GInv<T> g = contravariant_ref<KIND, T, SharedDivNoFixup>(bh, r, sin_theta, cos_theta);
            return rhs_ref_at<KIND, T, SharedDivNoFixup>(bh, r, sin_theta, cos_theta, g, p_t, p_r, p_th, p_ph);
        }
        const bool ok = bh.divs_ok && divs_ok_point(r, sin_theta, cos_theta);
        if (__ballot(!ok) == 0ull) {
            const GInv<T> g = contravariant_ref<KIND, T, SharedDiv>(bh, r, sin_theta, cos_theta);
            return rhs_ref_at<KIND, T, SharedDiv>(bh, r, sin_theta, cos_theta, g, p_t, p_r, p_th, p_ph);
        }
    }
    const GInv<T> g = contravariant_ref<KIND, T>(bh, r, sin_theta, cos_theta);
    return rhs_ref_at<KIND, T>(bh, r, sin_theta, cos_theta, g, p_t, p_r, p_th, p_ph);
}

// ---------------------------------------------------------------------------
// FAST Kerr-Schild right-hand side: the same Hamilton equations with one
// shared reciprocal 1/(Sigma sin^2) and everything factored over 1/Sigma^2.
// Algebraically identical to rhs_ref<KS>; differs by rounding only.
// The point-only part (KsGeom) is split off so the first stage of a try can
// reuse the geometry the post-step bookkeeping already evaluated there.
// ---------------------------------------------------------------------------
struct KsGeom {
    double sin2;   // max(sin^2, 1e-12)
    double sc;     // sin * cos
    double sigma;  // r^2 + a^2 cos^2
    double delta;  // r^2 - 2Mr + a^2
    double inv_ss; // 1 / (sigma sin^2)
    bool polar;    // |sin| < 1e-10  (dH/dtheta := 0, kerr.rs:494)
};

// one Horner step z p + c with the coefficient in a scalar register pair, as ONE v_fma_f64 (the same
// operation: bits unchanged).  Left to itself the compiler keeps the ten coefficients of the post-step
// evaluation in twenty vector registers and issues every step as v_mov_b64 + v_fmac_f64 (the two-address
// form overwrites its addend): ten moves per try and 16 VGPRs (168 -> 152) for nothing; c3 +0.7 %
// (profiles/r04_ab_horner_sgpr.jsonl)
__device__ __forceinline__ double horner_step(double z, double p, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(z), "v"(p), "s"(c));
    return r;
}

// the top step z c1 + c0: the leading coefficient lives in a vector register pair (loop-invariant), the
// other comes from scalar registers
__device__ __forceinline__ double horner_top(double z, double c1, double c0) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(z), "v"(c1), "s"(c0));
    return r;
}

// sin^2 and sin*cos are all the right-hand side needs, so the quadrant logic of a
// full sincos collapses to one swap and one sign.
__device__ __forceinline__ KsGeom ks_geom(const Hole<double> &bh, double r, double theta) {
#if GRV_TRIG_BITS
    // j = round-to-nearest-even(theta 2/pi) by adding 1.5 2^52: the sum keeps the integer in its low
    // mantissa bits (|theta| < 2^50), so the quadrant's parity is bit 0 of the sum's low word and needs
    // no conversion, compare or select further down
    constexpr double kRound = 6755399441055744.0;
    const double t = fma(theta, 6.36619772367581382433e-01, kRound); // 2/pi
    const double j = t - kRound;
#else
    const double j = rint(theta * 6.36619772367581382433e-01); // 2/pi
#endif
    // pi/2 = hi + mid to 2^-107: with fma the product j*hi is not rounded, so two terms
    // leave |j| * 1.5e-33 -- nothing for the O(1..100) theta of a geodesic
    double x = fma(-j, 1.57079632679489655800e+00, theta);
    x = fma(-j, 6.12323399573676603587e-17, x);
    const double z = x * x;
    double ps = horner_top(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = horner_step(z, ps, 2.75573137070700676789e-06);
    ps = horner_step(z, ps, -1.98412698298579493134e-04);
    ps = horner_step(z, ps, 8.33333333332248946124e-03);
    ps = horner_step(z, ps, -1.66666666666666324348e-01);
    const double sr = fma(x * z, ps, x);
    double pc = horner_top(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = horner_step(z, pc, -2.75573143513906633035e-07);
    pc = horner_step(z, pc, 2.48015872894767294178e-05);
    pc = horner_step(z, pc, -1.38888888888741095749e-03);
    pc = horner_step(z, pc, 4.16666666666666019037e-02);
    const double cr = fma(z, fma(z, pc, -0.5), 1.0);
    KsGeom g;
    const double prod = sr * cr;
#if GRV_TRIG_BITS
    // odd quadrant: sin <-> cos, product changes sign -- both done on the bit patterns: a bitwise
    // select under the sign-extended parity bit (v_bfe_i32 + v_bfi_b32 per word) and the parity bit
    // added into the product's sign position (v_lshl_add_u32); 3 instructions fewer per evaluation
    // than convert / and / compare / two selects / xor / select
    const int tl = __double2loint(t);
    const int odd = bits_sext_bit0(tl); // 0 or ~0
    const double sn = __hiloint2double(bits_select(odd, __double2hiint(cr), __double2hiint(sr)),
                                       bits_select(odd, __double2loint(cr), __double2loint(sr)));
    g.sc = __hiloint2double(bits_shl_add(tl, 31, __double2hiint(prod)), __double2loint(prod));
#else
    const bool odd = ((int)j & 1) != 0; // odd quadrant: sin <-> cos, product changes sign
    const double sn = odd ? cr : sr;
    g.sc = odd ? -prod : prod;
#endif
    const double s2 = sn * sn;
    g.polar = s2 < 1e-20;
    g.sin2 = fmax(s2, 1e-12);
    const double r2a2 = fma(r, r, bh.a2);
    g.sigma = fma(-bh.a2, g.sin2, r2a2); // r^2 + a^2 (1 - sin^2)
    g.delta = fma(-bh.two_m, r, r2a2);
    g.inv_ss = fast_rcp3(g.sigma * g.sin2);
    return g;
}

// per-ray products of the constants of motion, formed once per launch
struct KsRayConsts {
    double pt, pph;
    double pt2;       // p_t^2
    double m2pt;      // -2 p_t
    double pph2;      // p_phi^2
    double a_pph;     // a p_phi
    double two_a_pph; // 2 a p_phi
};
__device__ __forceinline__ KsRayConsts ks_ray_consts(const Hole<double> &bh, double p_t, double p_ph) {
    KsRayConsts c;
    c.pt = p_t;
    c.pph = p_ph;
    c.pt2 = p_t * p_t;
    c.m2pt = -2.0 * p_t;
    c.pph2 = p_ph * p_ph;
    c.a_pph = bh.a * p_ph;
    c.two_a_pph = 2.0 * c.a_pph;
    return c;
}

// `ham` (optional) receives H = N / (2 Sigma) = (W - Sigma p_t^2) / (2 Sigma) of the same point,
// i.e. invariants/mod.rs:25-37 from the bracket the forces already need.
__device__ __forceinline__ Deriv<double> rhs_ks_geom(const Hole<double> &bh, const KsGeom &g,
                                                     double r, const KsRayConsts &c, double p_r,
                                                     double p_th, double *ham = nullptr) {
    // With N = 2 Sigma H the two force terms share one bracket W = N + Sigma p_t^2:
    //   W = Delta p_r^2 + p_th^2 + p_ph^2/sin^2 + 2 a p_r p_ph - 2Mr (p_t^2 - 2 p_t p_r)
    //   Sigma^2 dH/dr  = Sigma ((r-M) p_r^2 - M (p_t^2 - 2 p_t p_r)) - r W
    //   Sigma^2 dH/dth = sin cos (a^2 W - Sigma p_ph^2 / sin^4)
    // (the Sigma p_t^2 pieces of N and of dN cancel identically).
    const double sigma = g.sigma, delta = g.delta, inv_ss = g.inv_ss;
    const double isig = inv_ss * g.sin2; // 1/Sigma
    const double isin2 = inv_ss * sigma; // 1/sin^2
    const double two_mr = bh.two_m * r;
    const double pt_mix = fma(c.m2pt, p_r, c.pt2); // p_t^2 - 2 p_t p_r

    Deriv<double> d;
    d.dt = fma(two_mr * isig, p_r - c.pt, -c.pt);                     // g^tt p_t + g^tr p_r
    d.dr = isig * fma(two_mr, c.pt, fma(delta, p_r, c.a_pph));        // g^tr p_t + g^rr p_r + g^rph p_ph
    d.dth = isig * p_th;
    d.dph = isig * fma(c.pph, isin2, bh.a * p_r);                     // g^rph p_r + g^phph p_ph

    const double q = c.pph2 * isin2; // p_ph^2 / sin^2
    const double pr2 = p_r * p_r;
    double w = fma(c.two_a_pph, p_r, -(two_mr * pt_mix));
    w = w + q;
    w = fma(p_th, p_th, w);
    w = fma(delta, pr2, w);

    // both halved: Sigma^2 dH/dr and Sigma^2 dH/dth
    const double ar_half = fma(-r, w, sigma * fma(r - bh.M, pr2, -(bh.M * pt_mix)));
    const double ath_half = g.sc * fma(bh.a2, w, -(sigma * (q * isin2)));
    const double isig2 = isig * isig;
    d.dpr = -(isig2 * ar_half);
#if GRV_POLAR_HI_WORD
    // kerr.rs:494 (dH/dtheta := 0 within 1e-10 of the axis): only the high word of the force is
    // cleared (one select instead of two).  What is left in the low word is a subnormal below 2^-1042:
    // added to any normal p_theta it is absorbed exactly, and a ray that sits on the axis with
    // p_theta = 0 keeps |p_theta| < 1e-300 -- the same trajectory to every digit the output carries.
    {
        const double f = -(isig2 * ath_half);
        d.dpth = __hiloint2double(g.polar ? 0 : __double2hiint(f), __double2loint(f));
    }
#else
    d.dpth = g.polar ? 0.0 : -(isig2 * ath_half);
#endif
    if (ham) *ham = (0.5 * isig) * fma(-sigma, c.pt2, w);
    return d;
}

template <typename T = double>
__device__ __forceinline__ Deriv<T> rhs_ks_fast(const Hole<T> &bh, T r, T theta, T p_t, T p_r,
                                                T p_th, T p_ph) {
    return rhs_ks_geom(bh, ks_geom(bh, r, theta), r, ks_ray_consts(bh, p_t, p_ph), p_r, p_th);
}

__device__ __forceinline__ GInv<double> ginv_from_geom(const Hole<double> &bh, const KsGeom &g,
                                                       double r) {
    const double isig = g.inv_ss * g.sin2;
    GInv<double> o;
    o.tr = 2.0 * bh.M * r * isig;
    o.tt = -(1.0 + o.tr);
    o.tph = 0.0;
    o.rr = g.delta * isig;
    o.thth = isig;
    o.phph = g.inv_ss;
    o.rph = bh.a * isig;
    return o;
}

template <int KIND, int ARITH, typename T>
__device__ __forceinline__ Deriv<T> rhs(const Hole<T> &bh, T r, T theta, T p_t, T p_r, T p_th,
                                        T p_ph, bool nf_ok = false) {
    if constexpr (ARITH == GRV_ARITH_FAST && KIND == GRV_METRIC_KERR_KS)
        return rhs_ks_fast<T>(bh, r, theta, p_t, p_r, p_th, p_ph);
    else
        return rhs_ref<KIND, T>(bh, r, theta, p_t, p_r, p_th, p_ph, nf_ok);
}

// ---------------------------------------------------------------------------
// hamiltonian (invariants/mod.rs:25-37) and renormalize_null
// (invariants/renormalization.rs:13-45) from one metric evaluation.
// `do_renorm` first projects p_r, then H is evaluated on the projected state,
// exactly as geodesic/mod.rs:229-237 does with two contravariant() calls.
// ---------------------------------------------------------------------------
// SKIP_TPH (Kerr-Schild under divs_nf_consts only): the partial sum ahead of the g^{t phi} term starts
// with the non-zero g^tt p_t^2, so it is never -0 and adding the term's +-0 leaves it as it is
template <int KIND, typename T, bool SKIP_TPH = false>
__device__ __forceinline__ T hamiltonian_of(const GInv<T> &g, T p_t, T p_r, T p_th, T p_ph) {
    // invariants/mod.rs:25-37, zero entries of the metric included (see rhs_ref)
    if constexpr (SKIP_TPH && KIND == GRV_METRIC_KERR_KS)
        return T(0.5) * (g.tt * p_t * p_t + g.rr * p_r * p_r + g.thth * p_th * p_th + g.phph * p_ph * p_ph +
                         T(2) * g.tr * p_t * p_r + T(2) * g.rph * p_r * p_ph);
    else
        return T(0.5) * (g.tt * p_t * p_t + g.rr * p_r * p_r + g.thth * p_th * p_th + g.phph * p_ph * p_ph +
                         T(2) * g.tph * p_t * p_ph + T(2) * g.tr * p_t * p_r + T(2) * g.rph * p_r * p_ph);
}

template <int KIND, int ARITH, typename T, bool SKIP_TPH = false>
__device__ __forceinline__ T renormalized_pr(const GInv<T> &g, T p_t, T p_r, T p_th, T p_ph) {
    const T a_quad = g.rr;
    // renormalization.rs:21-27.  STRICT keeps the zero entries of the metric (see rhs_ref); the
    // FAST contract, defined on finite states only, drops the Kerr-Schild g^{t phi} = 0 term
    const T b_quad = T(2) * (g.tr * p_t + g.rph * p_ph);
    T c_quad = g.tt * p_t * p_t + g.thth * p_th * p_th + g.phph * p_ph * p_ph;
    if constexpr (!(ARITH == GRV_ARITH_FAST && KIND == GRV_METRIC_KERR_KS) && !(SKIP_TPH && KIND == GRV_METRIC_KERR_KS))
        c_quad = c_quad + T(2) * g.tph * p_t * p_ph;
    T out = p_r;
    if (fabs_t(a_quad) > T(1e-12)) {
        const T disc = b_quad * b_quad - T(4) * a_quad * c_quad;
        if (disc >= T(0)) {
            const T sq = sqrt_t(disc);
            T sol1, sol2;
            if constexpr (ARITH == GRV_ARITH_FAST) {
                const T inv2a = fast_rcp(T(2) * a_quad);
                sol1 = (-b_quad + sq) * inv2a;
                sol2 = (-b_quad - sq) * inv2a;
            } else {
                sol1 = (-b_quad + sq) / (T(2) * a_quad);
                sol2 = (-b_quad - sq) / (T(2) * a_quad);
            }
            out = (fabs_t(sol1 - p_r) < fabs_t(sol2 - p_r)) ? sol1 : sol2;
        }
    }
    return out;
}

// metric at (r, theta) for the post-step bookkeeping
template <int KIND, int ARITH, typename T>
__device__ __forceinline__ GInv<T> contravariant_at(const Hole<T> &bh, T r, T theta) {
    if constexpr (ARITH == GRV_ARITH_FAST && KIND == GRV_METRIC_KERR_KS) {
        return ginv_from_geom(bh, ks_geom(bh, r, theta), r);
    } else {
        T s, c;
        sincos_t(theta, &s, &c);
        return contravariant_ref<KIND, T>(bh, r, s, c);
    }
}

} // namespace
