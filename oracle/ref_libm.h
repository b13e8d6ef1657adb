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
 * fdlibm / msun algorithms (k_sin, k_cos, medium-range rem_pio2, e_pow, e_exp, s_atan, e_log, e_acos, e_atan2): IEEE add, multiply,
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
double orc_log(double x);
double orc_acos(double x);
double orc_atan2(double y, double x);

/* f32 forms used by the shader-order f32 kernels and their oracle: the f64 routine, rounded once */
float orc_sinf(float x);
float orc_cosf(float x);
float orc_powf(float x, float y);
float orc_expf(float x);
float orc_logf(float x);
float orc_acosf(float x);
float orc_atan2f(float y, float x);

#ifdef __cplusplus
}
#endif
#endif
