/*
 * ref_libm.h -- the transcendental functions of the f64 ray path with a SPECIFIED result.
 * TEST INFRASTRUCTURE ONLY (part of the oracle).
 *
 * The reference calls Rust's f64::sin / cos / powf.  On its production target
 * (wasm32-unknown-unknown) those lower to the compiler's bundled port of the FreeBSD msun /
 * fdlibm routines; natively they lower to the platform libm.  Either way the last bit is the
 * libm's, not the reference's ("last-ulp differences are inherent").  For a parity
 * statement that does not depend on which libm happens to be linked, the oracle and the engine's
 * STRICT kernels both evaluate these functions by the routines restated here -- the published
 * fdlibm / msun algorithms (k_sin, k_cos, medium-range rem_pio2, e_pow, e_exp, s_atan): IEEE add, multiply,
 * divide and sqrt only, no FMA, so the result is a pure function of the argument on any IEEE-754
 * machine.  Accuracy is checked against mpmath in tests/test_ref_libm.py (< 1 ulp).
 *
 * Domain note: rem_pio2's Payne-Hanek branch (|x| >= 2^20 pi/2 ~ 1.6e6) is not restated; such
 * arguments are reduced by the same three-term Cody-Waite steps, which loses accuracy gradually
 * above that size.  The polar angle of a geodesic is O(1..100).
 */
#ifndef REF_LIBM_H
#define REF_LIBM_H

#ifdef __cplusplus
extern "C" {
#endif

double orc_sin(double x);
double orc_cos(double x);
double orc_pow(double x, double y);
double orc_exp(double x);
double orc_atan(double x);

#ifdef __cplusplus
}
#endif
#endif
