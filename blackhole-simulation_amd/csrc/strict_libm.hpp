// strict_libm.hpp -- sin, cos, pow, exp, log, atan, atan2 and acos with a specified result
// (f64, and f32 forms = the f64 routine rounded once), for the STRICT contract.
//
// The reference calls Rust's f64::sin / cos / powf; their last bit belongs to whichever libm the
// build links (the wasm target's bundled msun port, or the platform's).  So that a STRICT result
// is a pure function of its inputs -- the same bits from this engine on any device and from the
// checker in oracle/ -- the STRICT kernels evaluate them by the published fdlibm / FreeBSD msun
// algorithms written out here (k_sin, k_cos, the medium-range Cody-Waite rem_pio2, e_pow, e_exp, s_atan, e_log, e_acos, e_atan2): IEEE
// add / multiply / divide / sqrt only, round-to-nearest, no FMA (this header is only used from
// translation units built with -ffp-contract=off).  Errors against mpmath, tests/test_ref_libm.py:
// < 1 ulp for sin, cos, pow, exp, log, atan, acos (worst seen 0.86); atan2 up to 1.12 ulp (the
// quotient y/x is rounded before s_atan sees it, as in e_atan2.c -- kept, the result is specified).
// Not restated: rem_pio2's Payne-Hanek branch; |x| >= 2^20 pi/2 goes through the same
// three-term reduction and loses accuracy gradually (a geodesic's theta is O(1..100)).
//
// The algorithms, their polynomial coefficients and split constants are those of fdlibm (k_sin.c,
// k_cos.c, e_rem_pio2.c, e_pow.c, e_exp.c, e_log.c, s_atan.c, e_atan2.c, e_acos.c), whose files
// carry this notice:
//   ====================================================
//   Copyright (C) 1993, 2004 by Sun Microsystems, Inc. All rights reserved.
//
//   Developed at SunSoft / SunPro, a Sun Microsystems, Inc. business.
//   Permission to use, copy, modify, and distribute this
//   software is freely granted, provided that this notice
//   is preserved.
//   ====================================================
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#define GRV_HD __host__ __device__
// A/B switch: products that are exact by construction contracted into the neighbouring sum (same bits)
#ifndef GRV_STRICT_EXACT_FMA
#define GRV_STRICT_EXACT_FMA 1
#endif

namespace strictm {

GRV_HD inline uint64_t bits_of(double x) {
    uint64_t u;
    __builtin_memcpy(&u, &x, sizeof u);
    return u;
}
GRV_HD inline double from_bits(uint64_t u) {
    double x;
    __builtin_memcpy(&x, &u, sizeof x);
    return x;
}
GRV_HD inline int32_t hi_word(double x) { return (int32_t)(bits_of(x) >> 32); }
GRV_HD inline uint32_t lo_word(double x) { return (uint32_t)bits_of(x); }
GRV_HD inline double with_hi(double x, int32_t hi) {
    return from_bits(((uint64_t)(uint32_t)hi << 32) | (bits_of(x) & 0xffffffffull));
}
GRV_HD inline double clear_lo(double x) { return from_bits(bits_of(x) & 0xffffffff00000000ull); }

/* ---- k_sin / k_cos on [-pi/4, pi/4], argument x + y (y = tail) ---- */
constexpr double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                    S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                    S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
constexpr double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                    C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                    C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;

GRV_HD inline double k_sin(double x, double y, int iy) {
    const double z = x * x;
    const double w = z * z;
    const double r = S2 + z * (S3 + z * S4) + z * w * (S5 + z * S6);
    const double v = z * x;
    if (iy == 0) return x + v * (S1 + z * r);
    /* 0.5 * y is exact, so the fused form rounds 0.5 y - v r once, exactly as the separate product and
     * difference do: same bits, one instruction less */
#if GRV_STRICT_EXACT_FMA
    return x - ((z * __builtin_fma(0.5, y, -(v * r)) - y) - v * S1);
#else
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
#endif
}

GRV_HD inline double k_cos(double x, double y) {
    const double z = x * x;
    double w = z * z;
    const double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
    /* hz = 0.5 z is exact: w = 1 - hz and (1 - w) - hz through fma(-0.5, z, .) are the same
     * differences without forming hz (same bits, one instruction less) */
#if GRV_STRICT_EXACT_FMA
    w = __builtin_fma(-0.5, z, 1.0);
    return w + (__builtin_fma(-0.5, z, 1.0 - w) + (z * r - x * y));
#else
    const double hz = 0.5 * z;
    w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
#endif
}

/* ---- x = n pi/2 + (y0 + y1), |y0 + y1| <= pi/4: Cody-Waite with 33+33+33+53 bits of pi/2 ---- */
GRV_HD inline int rem_pio2_medium(double x, double *y0, double *y1) {
    constexpr double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                        pio2_1t = 6.07710050650619224932e-11, pio2_2 = 6.07710050630396597660e-11,
                        pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
                        pio2_3t = 8.47842766036889956997e-32;
    const double fn = ::rint(x * invpio2);
    double r = x - fn * pio2_1;
    double w = fn * pio2_1t; /* 1st round, good to 85 bits */
    double y = r - w;
    const int ex = (hi_word(x) >> 20) & 0x7ff;
    int ey = (hi_word(y) >> 20) & 0x7ff;
    if (ex - ey > 16) { /* 2nd round, good to 118 bits */
        double t = r;
        w = fn * pio2_2;
        r = t - w;
        w = fn * pio2_2t - ((t - r) - w);
        y = r - w;
        ey = (hi_word(y) >> 20) & 0x7ff;
        if (ex - ey > 49) { /* 3rd round, good to 151 bits */
            t = r;
            w = fn * pio2_3;
            r = t - w;
            w = fn * pio2_3t - ((t - r) - w);
            y = r - w;
        }
    }
    *y0 = y;
    *y1 = (r - y) - w;
    /* quadrant = fn mod 4, exact for any integer-valued fn */
#if GRV_STRICT_EXACT_FMA
    return (int)__builtin_fma(-4.0, ::floor(fn * 0.25), fn); /* 4 floor(.) is exact: the same difference */
#else
    return (int)(fn - 4.0 * ::floor(fn * 0.25));
#endif
}

GRV_HD inline double sl_sin(double x) {
    const int32_t ix = hi_word(x) & 0x7fffffff;
    if (ix <= 0x3fe921fb) { /* |x| <= pi/4 */
        if (ix < 0x3e500000) return x; /* |x| < 2^-26 */
        return k_sin(x, 0.0, 0);
    }
    if (ix >= 0x7ff00000) return x - x; /* inf, NaN */
    double y0, y1;
    const int n = rem_pio2_medium(x, &y0, &y1);
    switch (n & 3) {
    case 0: return k_sin(y0, y1, 1);
    case 1: return k_cos(y0, y1);
    case 2: return -k_sin(y0, y1, 1);
    default: return -k_cos(y0, y1);
    }
}

GRV_HD inline double sl_cos(double x) {
    const int32_t ix = hi_word(x) & 0x7fffffff;
    if (ix <= 0x3fe921fb) {
        if (ix < 0x3e46a09e) return 1.0; /* |x| < 2^-27 sqrt(2) */
        return k_cos(x, 0.0);
    }
    if (ix >= 0x7ff00000) return x - x;
    double y0, y1;
    const int n = rem_pio2_medium(x, &y0, &y1);
    switch (n & 3) {
    case 0: return k_cos(y0, y1);
    case 1: return -k_sin(y0, y1, 1);
    case 2: return -k_cos(y0, y1);
    default: return k_sin(y0, y1, 1);
    }
}

/* ---- x * 2^n without double rounding on the way into the subnormals ---- */
GRV_HD inline double scale2(double x, int n) {
    double y = x;
    if (n > 1023) {
        y *= 0x1p1023;
        n -= 1023;
        if (n > 1023) {
            y *= 0x1p1023;
            n -= 1023;
            if (n > 1023) n = 1023;
        }
    } else if (n < -1022) {
        y *= 0x1p-1022 * 0x1p53;
        n += 1022 - 53;
        if (n < -1022) {
            y *= 0x1p-1022 * 0x1p53;
            n += 1022 - 53;
            if (n < -1022) n = -1022;
        }
    }
    return y * from_bits((uint64_t)(0x3ff + n) << 52);
}

/* ---- pow: log2(x) in two pieces to ~ 2^-77, times y in two pieces, then 2^z ---- */
GRV_HD inline double sl_pow(double x, double y) {
    constexpr double dp_h1 = 5.84962487220764160156e-01, dp_l1 = 1.35003920212974897128e-08;
    constexpr double two53 = 9007199254740992.0, huge = 1.0e300, tiny = 1.0e-300;
    constexpr double L1 = 5.99999999999994648725e-01, L2 = 4.28571428578550184252e-01,
                        L3 = 3.33333329818377432918e-01, L4 = 2.72728123808534006489e-01,
                        L5 = 2.30660745775561754067e-01, L6 = 2.06975017800338417784e-01;
    constexpr double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                        P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                        P5 = 4.13813679705723846039e-08;
    constexpr double lg2 = 6.93147180559945286227e-01, lg2_h = 6.93147182464599609375e-01,
                        lg2_l = -1.90465429995776804525e-09, ovt = 8.0085662595372944372e-17,
                        cp = 9.61796693925975554329e-01, cp_h = 9.61796700954437255859e-01,
                        cp_l = -7.02846165095275826516e-09, ivln2 = 1.44269504088896338700e+00,
                        ivln2_h = 1.44269502162933349609e+00, ivln2_l = 1.92596299112661746887e-08;

    const int32_t hx = hi_word(x), hy = hi_word(y);
    const uint32_t lx = lo_word(x), ly = lo_word(y);
    int32_t ix = hx & 0x7fffffff;
    const int32_t iy = hy & 0x7fffffff;

    if (((uint32_t)iy | ly) == 0) return 1.0;          /* x^0 = 1 */
    if (hx == 0x3ff00000 && lx == 0) return 1.0;       /* 1^y = 1, even for NaN */
    if (ix > 0x7ff00000 || (ix == 0x7ff00000 && lx != 0) || iy > 0x7ff00000 ||
        (iy == 0x7ff00000 && ly != 0))
        return (x + 0.0) + (y + 0.0);                  /* NaN in, NaN out */

    /* for x < 0: 0 = y not an integer, 1 = odd integer, 2 = even integer */
    int yisint = 0;
    if (hx < 0) {
        if (iy >= 0x43400000)
            yisint = 2;
        else if (iy >= 0x3ff00000) {
            const int k = (iy >> 20) - 0x3ff;
            if (k > 20) {
                const uint32_t j = ly >> (52 - k);
                if ((j << (52 - k)) == ly) yisint = 2 - (int)(j & 1u);
            } else if (ly == 0) {
                const int32_t j = iy >> (20 - k);
                if ((j << (20 - k)) == iy) yisint = 2 - (j & 1);
            }
        }
    }

    if (ly == 0) { /* special exponents */
        if (iy == 0x7ff00000) {
            if ((((uint32_t)(ix - 0x3ff00000)) | lx) == 0) return 1.0; /* (-1)^+-inf */
            if (ix >= 0x3ff00000) return hy >= 0 ? y : 0.0;
            return hy < 0 ? -y : 0.0;
        }
        if (iy == 0x3ff00000) return hy < 0 ? 1.0 / x : x;
        if (hy == 0x40000000) return x * x;
        if (hy == 0x3fe00000 && hx >= 0) return ::sqrt(x);
    }

    double ax = ::fabs(x);
    if (lx == 0 && (ix == 0x7ff00000 || ix == 0 || ix == 0x3ff00000)) { /* x = +-0, +-inf, +-1 */
        double z = ax;
        if (hy < 0) z = 1.0 / z;
        if (hx < 0) {
            if (((ix - 0x3ff00000) | yisint) == 0)
                z = (z - z) / (z - z); /* (-1)^non-integer */
            else if (yisint == 1)
                z = -z;
        }
        return z;
    }

    const int neg = hx < 0;
    if (neg && yisint == 0) return (x - x) / (x - x); /* negative ^ non-integer */
    const double sgn = (neg && yisint == 1) ? -1.0 : 1.0;

    double t1, t2;
    if (iy > 0x41e00000) { /* |y| > 2^31 */
        if (iy > 0x43f00000) { /* |y| > 2^64: must over/underflow */
            if (ix <= 0x3fefffff) return hy < 0 ? huge * huge : tiny * tiny;
            if (ix >= 0x3ff00000) return hy > 0 ? huge * huge : tiny * tiny;
        }
        if (ix < 0x3fefffff) return hy < 0 ? sgn * huge * huge : sgn * tiny * tiny;
        if (ix > 0x3ff00000) return hy > 0 ? sgn * huge * huge : sgn * tiny * tiny;
        /* |1 - x| <= 2^-20: log(x) by x - x^2/2 + x^3/3 - x^4/4 */
        const double t = ax - 1.0;
        const double w = (t * t) * (0.5 - t * (0.3333333333333333333333 - t * 0.25));
        const double u = ivln2_h * t;
        const double v = t * ivln2_l - w * ivln2;
        t1 = clear_lo(u + v);
        t2 = v - (t1 - u);
    } else {
        int n = 0;
        if (ix < 0x00100000) { /* subnormal x */
            ax *= two53;
            n -= 53;
            ix = hi_word(ax);
        }
        n += (ix >> 20) - 0x3ff;
        const int32_t j = ix & 0x000fffff;
        int k;
        ix = j | 0x3ff00000;
        if (j <= 0x3988E)
            k = 0; /* |x| < sqrt(3/2) */
        else if (j < 0xBB67A)
            k = 1; /* |x| < sqrt(3) */
        else {
            k = 0;
            n += 1;
            ix -= 0x00100000;
        }
        ax = with_hi(ax, ix);

        /* ss = s_h + s_l = (x - bp) / (x + bp) */
        const double bpk = k ? 1.5 : 1.0, dp_hk = k ? dp_h1 : 0.0, dp_lk = k ? dp_l1 : 0.0;
        double u = ax - bpk;
        double v = 1.0 / (ax + bpk);
        const double ss = u * v;
        const double s_h = clear_lo(ss);
        double t_h = with_hi(0.0, ((ix >> 1) | 0x20000000) + 0x00080000 + (k << 18));
        double t_l = ax - (t_h - bpk);
        const double s_l = v * ((u - s_h * t_h) - s_h * t_l);
        /* log(ax) */
        double s2 = ss * ss;
        double r = s2 * s2 * (L1 + s2 * (L2 + s2 * (L3 + s2 * (L4 + s2 * (L5 + s2 * L6)))));
        r += s_l * (s_h + ss);
        s2 = s_h * s_h;
        t_h = clear_lo(3.0 + s2 + r);
        t_l = r - ((t_h - 3.0) - s2);
        u = s_h * t_h;
        v = s_l * t_h + t_l * ss;
        /* 2/(3 log 2) * (ss + ...) */
        const double p_h = clear_lo(u + v);
        const double p_l = v - (p_h - u);
        const double z_h = cp_h * p_h;
        const double z_l = cp_l * p_h + p_l * cp + dp_lk;
        /* log2(ax) = n + dp_h + z_h + z_l */
        const double t = (double)n;
        t1 = clear_lo(((z_h + z_l) + dp_hk) + t);
        t2 = z_l - (((t1 - t) - dp_hk) - z_h);
    }

    /* (y1 + y2) * (t1 + t2) */
    const double y1 = clear_lo(y);
    double p_l = (y - y1) * t1 + y * t2;
    double p_h = y1 * t1;
    double z = p_l + p_h;
    int32_t j = hi_word(z);
    const uint32_t i0 = lo_word(z);
    if (j >= 0x40900000) { /* z >= 1024 */
        if ((((uint32_t)(j - 0x40900000)) | i0) != 0) return sgn * huge * huge;
        if (p_l + ovt > z - p_h) return sgn * huge * huge;
    } else if ((j & 0x7fffffff) >= 0x4090cc00) { /* z <= -1075 */
        if ((((uint32_t)j - 0xc090cc00u) | i0) != 0) return sgn * tiny * tiny;
        if (p_l <= z - p_h) return sgn * tiny * tiny;
    }
    /* 2^(p_h + p_l) */
    const int32_t i = j & 0x7fffffff;
    int k = (i >> 20) - 0x3ff;
    int32_t n = 0;
    if (i > 0x3fe00000) { /* |z| > 0.5: n = [z + 0.5] */
        n = j + (0x00100000 >> (k + 1));
        k = ((n & 0x7fffffff) >> 20) - 0x3ff;
        const double t = with_hi(0.0, n & ~(0x000fffff >> k));
        n = ((n & 0x000fffff) | 0x00100000) >> (20 - k);
        if (j < 0) n = -n;
        p_h -= t;
    }
    double t = clear_lo(p_l + p_h);
    const double u = t * lg2_h;
    const double v = (p_l - (t - p_h)) * lg2 + t * lg2_l;
    z = u + v;
    const double w = v - (z - u);
    t = z * z;
    const double tt = z - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    const double r = (z * tt) / (tt - 2.0) - (w + z * w);
    z = 1.0 - (r - z);
    j = hi_word(z);
    j += (int32_t)((uint32_t)n << 20);
    if ((j >> 20) <= 0)
        z = scale2(z, n); /* subnormal result */
    else
        z = with_hi(z, j);
    return sgn * z;
}

/* ---- exp: x = k ln2 + r, |r| <= 0.5 ln2; exp(r) by the (r c)/(2 - c) rational form ---- */
GRV_HD inline double sl_exp(double x) {
    constexpr double o_threshold = 7.09782712893383973096e+02, u_threshold = -7.45133219101941108420e+02,
                        ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                        invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                        P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                        P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08,
                        huge = 1.0e+300, twom1000 = 9.33263618503218878990e-302; /* 2^-1000 */
    uint32_t hx = (uint32_t)hi_word(x);
    const int xsb = (int)(hx >> 31);
    hx &= 0x7fffffffu;
    double hi = 0.0, lo = 0.0;
    int k = 0;
    if (hx >= 0x40862E42u) { /* |x| >= 709.78 */
        if (hx >= 0x7ff00000u) {
            if (((hx & 0xfffffu) | lo_word(x)) != 0) return x + x; /* NaN */
            return xsb == 0 ? x : 0.0;                            /* exp(+-inf) = inf, 0 */
        }
        if (x > o_threshold) return huge * huge;
        if (x < u_threshold) return twom1000 * twom1000;
    }
    if (hx > 0x3fd62e42u) {     /* |x| > 0.5 ln2 */
        if (hx < 0x3FF0A2B2u) { /* and |x| < 1.5 ln2 */
            hi = xsb ? x + ln2HI : x - ln2HI;
            lo = xsb ? -ln2LO : ln2LO;
            k = 1 - xsb - xsb;
        } else {
            k = (int)(invln2 * x + (xsb ? -0.5 : 0.5));
            const double t = (double)k;
            hi = x - t * ln2HI;
            lo = t * ln2LO;
        }
        x = hi - lo;
    } else if (hx < 0x3e300000u) { /* |x| < 2^-28 */
        return 1.0 + x;
    }
    const double t = x * x;
    const double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    const double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    if (k >= -1021) return with_hi(y, hi_word(y) + (int32_t)((uint32_t)k << 20));
    return with_hi(y, hi_word(y) + (int32_t)((uint32_t)(k + 1000) << 20)) * twom1000;
}

/* ---- atan: argument folded onto [0, 7/16] around 0, 1/2, 1, 3/2, inf; odd/even split series ---- */
GRV_HD inline double sl_atan(double x) {
    constexpr double atanhi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01,
                                     9.82793723247329054082e-01, 1.57079632679489655800e+00};
    constexpr double atanlo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17,
                                     1.39033110312309984516e-17, 6.12323399573676603587e-17};
    constexpr double aT[11] = {3.33333333333329318027e-01,  -1.99999999998764832476e-01,
                                  1.42857142725034663711e-01,  -1.11111104054623557880e-01,
                                  9.09088713343650656196e-02,  -7.69187620504482999495e-02,
                                  6.66107313738753120669e-02,  -5.83357013379057348645e-02,
                                  4.97687799461593236017e-02,  -3.65315727442169155270e-02,
                                  1.62858201153657823623e-02};
    const int32_t hx = hi_word(x);
    const int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x44100000) { /* |x| >= 2^66 */
        if (ix > 0x7ff00000 || (ix == 0x7ff00000 && lo_word(x) != 0)) return x + x; /* NaN */
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3fdc0000) {            /* |x| < 0.4375 */
        if (ix < 0x3e200000) return x; /* |x| < 2^-29 */
        id = -1;
    } else {
        x = ::fabs(x);
        if (ix < 0x3ff30000) {     /* |x| < 1.1875 */
            if (ix < 0x3fe60000) { /* 7/16 <= |x| < 11/16 */
                id = 0;
                x = (2.0 * x - 1.0) / (2.0 + x);
            } else { /* 11/16 <= |x| < 19/16 */
                id = 1;
                x = (x - 1.0) / (x + 1.0);
            }
        } else if (ix < 0x40038000) { /* |x| < 2.4375 */
            id = 2;
            x = (x - 1.5) / (1.0 + 1.5 * x);
        } else { /* 2.4375 <= |x| < 2^66 */
            id = 3;
            x = -1.0 / x;
        }
    }
    const double z = x * x;
    const double w = z * z;
    const double s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const double s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    const double hi_c = id == 0 ? atanhi[0] : (id == 1 ? atanhi[1] : (id == 2 ? atanhi[2] : atanhi[3]));
    const double lo_c = id == 0 ? atanlo[0] : (id == 1 ? atanlo[1] : (id == 2 ? atanlo[2] : atanlo[3]));
    const double r = hi_c - ((x * (s1 + s2) - lo_c) - x);
    return hx < 0 ? -r : r;
}

/* ---- log: x = 2^k (1 + f), log(1 + f) = f - hfsq + s (hfsq + R(s^2)), s = f / (2 + f) ---- */
GRV_HD inline double sl_log(double x) {
    constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                        two54 = 1.80143985094819840000e+16, Lg1 = 6.666666666666735130e-01,
                        Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                        Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01,
                        Lg6 = 1.531383769920937332e-01, Lg7 = 1.479819860511658591e-01;
    int32_t hx = hi_word(x);
    const uint32_t lx = lo_word(x);
    int k = 0;
    if (hx < 0x00100000) { /* x < 2^-1022 */
        if ((((uint32_t)hx & 0x7fffffffu) | lx) == 0) return from_bits(0xfff0000000000000ull); /* log(+-0) = -inf */
        if (hx < 0) return (x - x) / (x - x);                               /* log(-#) = NaN */
        k -= 54;
        x *= two54; /* subnormal: scale up */
        hx = hi_word(x);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int32_t i = (hx + 0x95f64) & 0x100000;
    x = with_hi(x, hx | (i ^ 0x3ff00000)); /* normalize x or x/2 */
    k += (i >> 20);
    const double f = x - 1.0;
    const double dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) { /* -2^-20 <= f < 2^-20 */
        if (f == 0.0) {
            if (k == 0) return 0.0;
            return dk * ln2_hi + dk * ln2_lo;
        }
        const double R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    const double s = f / (2.0 + f);
    const double z = s * s;
    const int32_t ii = hx - 0x6147a;
    const double w = z * z;
    const int32_t jj = 0x6b851 - hx;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    if ((ii | jj) > 0) {
        const double hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    if (k == 0) return f - s * (f - R);
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

/* ---- acos: rational R(x^2) on |x| < 1/2, sqrt-based identities beyond ---- */
GRV_HD inline double sl_acos(double x) {
    constexpr double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
                        pi = 3.14159265358979311600e+00, pS0 = 1.66666666666666657415e-01,
                        pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                        pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04,
                        pS5 = 3.47933107596021167570e-05, qS1 = -2.40339491173441421878e+00,
                        qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                        qS4 = 7.70381505559019352791e-02;
    const int32_t hx = hi_word(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix >= 0x3ff00000) { /* |x| >= 1 */
        if ((((uint32_t)(ix - 0x3ff00000)) | lo_word(x)) == 0) return hx > 0 ? 0.0 : pi + 2.0 * pio2_lo;
        return (x - x) / (x - x); /* |x| > 1: NaN */
    }
    if (ix < 0x3fe00000) { /* |x| < 0.5 */
        if (ix <= 0x3c600000) return pio2_hi + pio2_lo; /* |x| < 2^-57 */
        const double z = x * x;
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (hx < 0) { /* x < -0.5 */
        const double z = (1.0 + x) * 0.5;
        const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const double s = ::sqrt(z);
        const double r = p / q;
        const double w = r * s - pio2_lo;
        return pi - 2.0 * (s + w);
    }
    /* x > 0.5 */
    const double z = (1.0 - x) * 0.5;
    const double s = ::sqrt(z);
    const double df = clear_lo(s);
    const double c = (z - df * df) / (s + df);
    const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const double r = p / q;
    const double w = r * s + c;
    return 2.0 * (df + w);
}

/* ---- atan2: quadrant logic around atan(|y / x|) ---- */
GRV_HD inline double sl_atan2(double y, double x) {
    constexpr double pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
    const int32_t hx = hi_word(x), hy = hi_word(y);
    const uint32_t lx = lo_word(x), ly = lo_word(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7ff00000 || (ix == 0x7ff00000 && lx != 0) || iy > 0x7ff00000 || (iy == 0x7ff00000 && ly != 0))
        return x + y; /* NaN */
    if ((((uint32_t)hx - 0x3ff00000u) | lx) == 0) return sl_atan(y); /* x = 1 */
    const int m = (int)(((uint32_t)hy >> 31) & 1u) | (int)(((uint32_t)hx >> 30) & 2u); /* 2 sign(x) + sign(y) */
    if (((uint32_t)iy | ly) == 0) { /* y = 0 */
        switch (m) {
        case 0:
        case 1: return y;  /* atan(+-0, +anything) = +-0 */
        case 2: return pi; /* atan(+0, -anything) = pi */
        default: return -pi;
        }
    }
    if (((uint32_t)ix | lx) == 0) return hy < 0 ? -pi / 2.0 : pi / 2.0; /* x = 0 */
    if (ix == 0x7ff00000) { /* x = inf */
        if (iy == 0x7ff00000) {
            switch (m) {
            case 0: return pi / 4.0;
            case 1: return -pi / 4.0;
            case 2: return 3.0 * pi / 4.0;
            default: return -3.0 * pi / 4.0;
            }
        }
        switch (m) {
        case 0: return 0.0;
        case 1: return -0.0;
        case 2: return pi;
        default: return -pi;
        }
    }
    if (iy == 0x7ff00000) return hy < 0 ? -pi / 2.0 : pi / 2.0; /* y = inf */
    double z;
    const int k = (iy - ix) >> 20;
    if (k > 60) { /* |y / x| > 2^60 */
        z = pi / 2.0 + 0.5 * pi_lo;
    } else if (hx < 0 && k < -60) {
        z = 0.0; /* 0 > |y| / x > -2^-60 */
    } else {
        z = sl_atan(::fabs(y / x));
    }
    switch (m) {
    case 0: return z;   /* atan(+, +) */
    case 1: return -z;  /* atan(-, +) */
    case 2: return pi - (z - pi_lo); /* atan(+, -) */
    default: return (z - pi_lo) - pi; /* atan(-, -) */
    }
}

/* ---- f32 forms: the f64 routine rounded once to f32 (what the shader-order f32 kernels use) ---- */
GRV_HD inline float sl_sinf(float x) { return (float)sl_sin((double)x); }
GRV_HD inline float sl_cosf(float x) { return (float)sl_cos((double)x); }
GRV_HD inline float sl_powf(float x, float y) { return (float)sl_pow((double)x, (double)y); }
GRV_HD inline float sl_expf(float x) { return (float)sl_exp((double)x); }
GRV_HD inline float sl_logf(float x) { return (float)sl_log((double)x); }
GRV_HD inline float sl_acosf(float x) { return (float)sl_acos((double)x); }
GRV_HD inline float sl_atan2f(float y, float x) { return (float)sl_atan2((double)y, (double)x); }

// both at once (one argument reduction)
GRV_HD inline void sl_sincos(double x, double *sn, double *cs) {
    const int32_t ix = hi_word(x) & 0x7fffffff;
    if (ix <= 0x3fe921fb) {
        *sn = (ix < 0x3e500000) ? x : k_sin(x, 0.0, 0);
        *cs = (ix < 0x3e46a09e) ? 1.0 : k_cos(x, 0.0);
        return;
    }
    if (ix >= 0x7ff00000) {
        *sn = *cs = x - x;
        return;
    }
    double y0, y1;
    const int n = rem_pio2_medium(x, &y0, &y1);
    const double s = k_sin(y0, y1, 1), c = k_cos(y0, y1);
    /* (a branch-free form -- two selects and two sign xors -- was measured in round 3: no faster, the
     * quadrant is nearly always uniform across a wave and the untaken cases are skipped) */
    switch (n & 3) {
    case 0: *sn = s; *cs = c; break;
    case 1: *sn = c; *cs = -s; break;
    case 2: *sn = -s; *cs = -c; break;
    default: *sn = -c; *cs = s; break;
    }
}

} // namespace strictm
#undef GRV_HD
