/*
 * gravitas_oracle.c -- CPU restatement (plain C, f64) of gravitas-core's
 * geodesic path.  TEST INFRASTRUCTURE ONLY -- see gravitas_oracle.h.
 *
 * Endpoints are "parity unpinned" by the reference's own tests (no reference
 * test runs integrate()); closed forms are pinned (tests/test_oracle_pins.py).
 *
 * Citations are file:line under /root/reference/physics-engine/gravitas-core/src
 * unless a crate is named.  Operation order follows the Rust expressions
 * (left-associative, same grouping); compile with -ffp-contract=off.
 *
 * Rust std semantics restated here:
 *   f64::max/min  -> fmax/fmin (NaN-ignoring)
 *   f64::clamp    -> rs_clamp  (x<lo?lo : x>hi?hi : x ; NaN passes through)
 *   f64::powi(n)  -> compiler-rt __powidf2 square-and-multiply (rs_powi)
 *   f64::powf, f64::sin, f64::cos on the per-ray path (metric, Hamilton derivatives, step
 *                    controller, g-factor, LUT axes) -> orc_pow / orc_sin / orc_cos of ref_libm.c:
 *                    written-out fdlibm-lineage routines with a specified result (the reference
 *                    takes these from whichever libm its target links; see ref_libm.h).  Closed
 *                    forms evaluated once per engine (ISCO, photon sphere ...) use the host libm,
 *                    as the engine's host code does.
 *   f64::signum   -> +1 for +0.0/positive, -1 for -0.0/negative, NaN for NaN
 */
#include "gravitas_oracle.h"
#include "ref_libm.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------- helpers */

static double rs_clamp(double x, double lo, double hi) {
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}

static double rs_powi(double a, int b) {
    /* compiler-rt __powidf2 */
    int recip = b < 0;
    double r = 1.0;
    while (1) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0 / r : r;
}

static double rs_signum(double x) {
    if (isnan(x)) return x;
    return signbit(x) ? -1.0 : 1.0;
}

/* ----------------------------------------------------------- construction */

orc_metric orc_metric_make(int kind, double mass, double spin) {
    orc_metric m;
    m.kind = kind;
    m.mass = mass;
    /* metric/kerr.rs:48-63: spin.clamp(-1.0, 1.0); Schwarzschild::spin() = 0
     * (metric/schwarzschild.rs:117-119) */
    m.spin = (kind == ORC_SCHWARZSCHILD) ? 0.0 : rs_clamp(spin, -1.0, 1.0);
    return m;
}

/* geodesic/integrator.rs:35-47 */
orc_options orc_options_default(void) {
    orc_options o;
    o.method = ORC_METHOD_RKF45;
    o.tolerance = 1e-8;
    o.initial_step = 0.01;
    o.max_steps = 10000;
    o.escape_radius = 1000.0;
    o.renormalize_interval = 10;
    o.step_size = 0.0;
    return o;
}

/* metric/kerr.rs:70-74 */
static double kerr_a(const orc_metric *m) { return m->spin * m->mass; }

/* ------------------------------------------------------------ closed forms */

/* metric/mod.rs:75-84 */
double orc_event_horizon(const orc_metric *mt) {
    double m = mt->mass;
    double a = mt->spin * m;
    double disc = m * m - a * a;
    if (disc < 0.0) return m;
    return m + sqrt(disc);
}

/* metric/kerr.rs:77-86 */
double orc_cauchy_horizon(const orc_metric *mt) {
    double m = mt->mass;
    double a = kerr_a(mt);
    double disc = m * m - a * a;
    if (disc < 0.0) return 0.0;
    return m - sqrt(disc);
}

/* metric/kerr.rs:91-94 ; Schwarzschild: metric/schwarzschild.rs:41-43 */
double orc_photon_sphere(const orc_metric *mt) {
    if (mt->kind == ORC_SCHWARZSCHILD) return 3.0 * mt->mass;
    double term = (2.0 / 3.0) * orc_acos(-mt->spin);
    return 2.0 * mt->mass * (1.0 + orc_cos(term));
}

/* metric/kerr.rs:100-123 ; Schwarzschild: metric/schwarzschild.rs:36-38 */
double orc_isco(const orc_metric *mt, int retrograde) {
    if (mt->kind == ORC_SCHWARZSCHILD) return 6.0 * mt->mass;
    double a_star = mt->spin;
    double m = mt->mass;
    if (fabs(a_star) < 1e-6) return m * 6.0;
    double a2 = a_star * a_star;
    double z1 = 1.0 + orc_pow(1.0 - a2, 1.0 / 3.0) *
                          (orc_pow(1.0 + a_star, 1.0 / 3.0) + orc_pow(1.0 - a_star, 1.0 / 3.0));
    double z2 = sqrt(3.0 * a2 + z1 * z1);
    double sign = retrograde ? 1.0 : -1.0;
    double disc = (3.0 - z1) * (3.0 + z1 + 2.0 * z2);
    double root = (disc < 0.0) ? 0.0 : sqrt(disc);
    return m * (3.0 + z2 + sign * root);
}

/* metric/kerr.rs:157-167 */
double orc_ergosphere(const orc_metric *mt, double theta) {
    double m = mt->mass;
    double a = kerr_a(mt);
    double cos_theta = orc_cos(theta);
    double disc = m * m - a * a * cos_theta * cos_theta;
    if (disc < 0.0) return m;
    return m + sqrt(disc);
}

/* metric/kerr.rs:172-176 */
double orc_keplerian_frequency(const orc_metric *mt, double r) {
    double m = mt->mass;
    double a = kerr_a(mt);
    return sqrt(m) / (orc_pow(r, 1.5) + a * sqrt(m));
}

/* ----------------------------------------------------------------- metric */

/* metric/kerr.rs:241-264 */
static void covariant_bl(const orc_metric *mt, double r, double theta, double g[16]) {
    double m = mt->mass;
    double a = kerr_a(mt);
    double r2 = r * r;
    double a2 = a * a;
    double sin_theta = orc_sin(theta);
    double cos_theta = orc_cos(theta);
    double sin2 = sin_theta * sin_theta;
    double cos2 = cos_theta * cos_theta;
    double sigma = r2 + a2 * cos2;
    double delta = r2 - 2.0 * m * r + a2;
    double g_tt = -(1.0 - (2.0 * m * r) / sigma);
    double g_rr = sigma / delta;
    double g_thth = sigma;
    double g_phph = (r2 + a2 + (2.0 * m * r * a2 * sin2) / sigma) * sin2;
    double g_tph = -(2.0 * m * r * a * sin2) / sigma;
    memset(g, 0, 16 * sizeof(double));
    g[0] = g_tt;
    g[3] = g_tph;
    g[5] = g_rr;
    g[10] = g_thth;
    g[12] = g_tph;
    g[15] = g_phph;
}

/* metric/kerr.rs:266-293 */
static void contravariant_bl(const orc_metric *mt, double r, double theta, double g[16]) {
    double m = mt->mass;
    double a = kerr_a(mt);
    double r2 = r * r;
    double a2 = a * a;
    double sin_theta = orc_sin(theta);
    double cos_theta = orc_cos(theta);
    double sin2 = sin_theta * sin_theta;
    double cos2 = cos_theta * cos_theta;
    double sigma = r2 + a2 * cos2;
    double delta = r2 - 2.0 * m * r + a2;
    double g_tt = -((sigma * (r2 + a2) + 2.0 * m * r * a2 * sin2) / (delta * sigma));
    double g_rr = delta / sigma;
    double g_thth = 1.0 / sigma;
    double g_phph = (sin2 < 1e-9) ? 0.0 : (delta - a2 * sin2) / (delta * sigma * sin2);
    double g_tph = -(2.0 * m * r * a) / (delta * sigma);
    memset(g, 0, 16 * sizeof(double));
    g[0] = g_tt;
    g[3] = g_tph;
    g[5] = g_rr;
    g[10] = g_thth;
    g[12] = g_tph;
    g[15] = g_phph;
}

/* metric/kerr.rs:295-372 */
static void hamiltonian_derivs_bl(const orc_metric *mt, double r, double theta, const double p[4],
                                  double *out_dr, double *out_dth) {
    double m = mt->mass;
    double a = kerr_a(mt);
    double r2 = r * r;
    double a2 = a * a;
    double cos_theta = orc_cos(theta);
    double sin_theta = orc_sin(theta);
    double sin2 = sin_theta * sin_theta;
    double cos2 = cos_theta * cos_theta;

    double sigma = r2 + a2 * cos2;
    double delta = r2 - 2.0 * m * r + a2;
    double sigma_sq = sigma * sigma;

    double dsigma_dr = 2.0 * r;
    double dsigma_dtheta = -2.0 * a2 * cos_theta * sin_theta;
    double ddelta_dr = 2.0 * r - 2.0 * m;

    double dg_rr_dr = (ddelta_dr * sigma - delta * dsigma_dr) / sigma_sq;
    double dg_rr_dtheta = -(delta * dsigma_dtheta) / sigma_sq;

    double dg_thth_dr = -dsigma_dr / sigma_sq;
    double dg_thth_dtheta = -dsigma_dtheta / sigma_sq;

    double num_tphi = -2.0 * m * r * a;
    double den_tphi = delta * sigma;
    double dnum_tphi_dr = -2.0 * m * a;
    double dden_tphi_dr = ddelta_dr * sigma + delta * dsigma_dr;
    double dg_tphi_dr =
        (dnum_tphi_dr * den_tphi - num_tphi * dden_tphi_dr) / (den_tphi * den_tphi);
    double dden_tphi_dtheta = delta * dsigma_dtheta;
    double dg_tphi_dtheta = -(num_tphi * dden_tphi_dtheta) / (den_tphi * den_tphi);

    double du_dr = dsigma_dr * (r2 + a2) + sigma * 2.0 * r + 2.0 * m * a2 * sin2;
    double dv_dr = dden_tphi_dr;
    double u_val = sigma * (r2 + a2) + 2.0 * m * r * a2 * sin2;
    double dg_tt_dr = -(du_dr * den_tphi - u_val * dv_dr) / (den_tphi * den_tphi);

    double du_dtheta =
        dsigma_dtheta * (r2 + a2) + 2.0 * m * r * a2 * 2.0 * sin_theta * cos_theta;
    double dv_dtheta = dden_tphi_dtheta;
    double dg_tt_dtheta = -(du_dtheta * den_tphi - u_val * dv_dtheta) / (den_tphi * den_tphi);

    double da_dr = -dsigma_dr / (sigma_sq * sin2);
    double db_dr = -a2 * dden_tphi_dr / (den_tphi * den_tphi);
    double dg_phph_dr = da_dr - db_dr;

    double d_denom_a_dtheta = dsigma_dtheta * sin2 + sigma * 2.0 * sin_theta * cos_theta;
    double da_dtheta = -d_denom_a_dtheta / (sigma_sq * sin2 * sin2);
    double db_dtheta = -a2 * dden_tphi_dtheta / (den_tphi * den_tphi);
    double dg_phph_dtheta = da_dtheta - db_dtheta;

    double p_t = p[0], p_r = p[1], p_th = p[2], p_ph = p[3];

    *out_dr = 0.5 * (p_t * p_t * dg_tt_dr + p_r * p_r * dg_rr_dr + p_th * p_th * dg_thth_dr +
                     p_ph * p_ph * dg_phph_dr + 2.0 * p_t * p_ph * dg_tphi_dr);
    *out_dth =
        0.5 * (p_t * p_t * dg_tt_dtheta + p_r * p_r * dg_rr_dtheta + p_th * p_th * dg_thth_dtheta +
               p_ph * p_ph * dg_phph_dtheta + 2.0 * p_t * p_ph * dg_tphi_dtheta);
}

/* metric/kerr.rs:379-410 */
static void covariant_ks(const orc_metric *mt, double r, double theta, double g[16]) {
    double m = mt->mass;
    double a = kerr_a(mt);
    double r2 = r * r;
    double a2 = a * a;
    double cos2 = rs_powi(orc_cos(theta), 2);
    double sin2 = 1.0 - cos2;
    double sigma = r2 + a2 * cos2;
    double h = (m * r) / sigma;
    double l_r = sigma / (r2 + a2);
    double l[4] = {1.0, l_r, 0.0, -a * sin2};
    double eta_tt = -1.0;
    double eta_rr = sigma / (r2 + a2);
    double eta_thth = sigma;
    double eta_phph = (r2 + a2) * sin2;
    memset(g, 0, 16 * sizeof(double));
    g[0] = eta_tt + 2.0 * h * l[0] * l[0];
    g[1] = 2.0 * h * l[0] * l[1];
    g[3] = 2.0 * h * l[0] * l[3];
    g[4] = 2.0 * h * l[1] * l[0];
    g[5] = eta_rr + 2.0 * h * l[1] * l[1];
    g[7] = 2.0 * h * l[1] * l[3];
    g[10] = eta_thth;
    g[12] = 2.0 * h * l[3] * l[0];
    g[13] = 2.0 * h * l[3] * l[1];
    g[15] = eta_phph + 2.0 * h * l[3] * l[3];
}

/* metric/kerr.rs:412-440 */
static void contravariant_ks(const orc_metric *mt, double r, double theta, double g[16]) {
    double m = mt->mass;
    double a = kerr_a(mt);
    double r2 = r * r;
    double a2 = a * a;
    double sin2 = fmax(rs_powi(orc_sin(theta), 2), 1e-12);
    double cos2 = 1.0 - sin2;
    double sigma = r2 + a2 * cos2;
    double delta = r2 - 2.0 * m * r + a2;
    double g_tt = -(1.0 + 2.0 * m * r / sigma);
    double g_tr = 2.0 * m * r / sigma;
    double g_rr = delta / sigma;
    double g_thth = 1.0 / sigma;
    double g_phph = 1.0 / (sigma * sin2);
    double g_rph = a / sigma;
    memset(g, 0, 16 * sizeof(double));
    g[0] = g_tt;
    g[1] = g_tr;
    g[4] = g_tr;
    g[5] = g_rr;
    g[7] = g_rph;
    g[10] = g_thth;
    g[13] = g_rph;
    g[15] = g_phph;
}

/* metric/kerr.rs:442-499 */
static void hamiltonian_derivs_ks(const orc_metric *mt, double r, double theta, const double p[4],
                                  double *out_dr, double *out_dth) {
    double m = mt->mass;
    double a = kerr_a(mt);
    double r2 = r * r;
    double a2 = a * a;
    double sin_theta = orc_sin(theta);
    double cos_theta = orc_cos(theta);
    double sin2 = fmax(sin_theta * sin_theta, 1e-12);
    double cos2 = 1.0 - sin2;
    double sigma = r2 + a2 * cos2;
    double sigma2 = sigma * sigma;
    double delta = r2 - 2.0 * m * r + a2;

    double dsigma_dr = 2.0 * r;
    double dsigma_dtheta = -2.0 * a2 * sin_theta * cos_theta;
    double ddelta_dr = 2.0 * r - 2.0 * m;

    double dg_tt_dr = -(2.0 * m * (sigma - r * dsigma_dr)) / sigma2;
    double dg_tt_dtheta = (2.0 * m * r * dsigma_dtheta) / sigma2;

    double dg_tr_dr = -dg_tt_dr;
    double dg_tr_dtheta = -dg_tt_dtheta;

    double dg_rr_dr = (ddelta_dr * sigma - delta * dsigma_dr) / sigma2;
    double dg_rr_dtheta = -(delta * dsigma_dtheta) / sigma2;

    double dg_thth_dr = -dsigma_dr / sigma2;
    double dg_thth_dtheta = -dsigma_dtheta / sigma2;

    double dg_phph_dr = -dsigma_dr / (sigma2 * sin2);
    double dg_phph_dtheta =
        -(dsigma_dtheta * sin2 + sigma * 2.0 * sin_theta * cos_theta) / (sigma2 * sin2 * sin2);

    double dg_rph_dr = -(a * dsigma_dr) / sigma2;
    double dg_rph_dtheta = -(a * dsigma_dtheta) / sigma2;

    double dh_dr = 0.5 * (dg_tt_dr * p[0] * p[0] + dg_rr_dr * p[1] * p[1] +
                          dg_thth_dr * p[2] * p[2] + dg_phph_dr * p[3] * p[3] +
                          2.0 * dg_tr_dr * p[0] * p[1] + 2.0 * dg_rph_dr * p[1] * p[3]);

    double dh_dtheta = 0.5 * (dg_tt_dtheta * p[0] * p[0] + dg_rr_dtheta * p[1] * p[1] +
                              dg_thth_dtheta * p[2] * p[2] + dg_phph_dtheta * p[3] * p[3] +
                              2.0 * dg_tr_dtheta * p[0] * p[1] + 2.0 * dg_rph_dtheta * p[1] * p[3]);

    if (fabs(sin_theta) < 1e-10) dh_dtheta = 0.0;

    *out_dr = dh_dr;
    *out_dth = dh_dtheta;
}

/* metric/schwarzschild.rs:47-60 */
static void covariant_schw(const orc_metric *mt, double r, double theta, double g[16]) {
    double m = mt->mass;
    double rs = 2.0 * m;
    double sin2 = rs_powi(orc_sin(theta), 2);
    memset(g, 0, 16 * sizeof(double));
    g[0] = -(1.0 - rs / r);
    g[5] = 1.0 / (1.0 - rs / r);
    g[10] = r * r;
    g[15] = r * r * sin2;
}

/* metric/schwarzschild.rs:62-75 */
static void contravariant_schw(const orc_metric *mt, double r, double theta, double g[16]) {
    double m = mt->mass;
    double rs = 2.0 * m;
    double sin2 = fmax(rs_powi(orc_sin(theta), 2), 1e-12);
    memset(g, 0, 16 * sizeof(double));
    g[0] = -1.0 / (1.0 - rs / r);
    g[5] = 1.0 - rs / r;
    g[10] = 1.0 / (r * r);
    g[15] = 1.0 / (r * r * sin2);
}

/* metric/schwarzschild.rs:77-111 */
static void hamiltonian_derivs_schw(const orc_metric *mt, double r, double theta,
                                    const double p[4], double *out_dr, double *out_dth) {
    double m = mt->mass;
    double r2 = r * r;
    double r3 = r2 * r;
    double sin_theta = orc_sin(theta);
    double cos_theta = orc_cos(theta);
    double sin2 = sin_theta * sin_theta;

    double f = 1.0 - 2.0 * m / r;
    double dg_tt_dr = -2.0 * m / (r2 * f * f);
    double dg_rr_dr = 2.0 * m / r2;
    double dg_thth_dr = -2.0 / r3;
    double dg_phph_dr = (sin2 < 1e-12) ? 0.0 : -2.0 / (r3 * sin2);
    double dg_phph_dtheta = (sin2 < 1e-12) ? 0.0 : -2.0 * cos_theta / (r2 * sin_theta * sin2);

    *out_dr = 0.5 * (dg_tt_dr * p[0] * p[0] + dg_rr_dr * p[1] * p[1] + dg_thth_dr * p[2] * p[2] +
                     dg_phph_dr * p[3] * p[3]);
    *out_dth = 0.5 * dg_phph_dtheta * p[3] * p[3];
}

/* metric/kerr.rs:207-226 dispatch */
void orc_covariant(const orc_metric *m, double r, double theta, double g[16]) {
    switch (m->kind) {
    case ORC_KERR_BL: covariant_bl(m, r, theta, g); break;
    case ORC_KERR_KS: covariant_ks(m, r, theta, g); break;
    default: covariant_schw(m, r, theta, g); break;
    }
}

void orc_contravariant(const orc_metric *m, double r, double theta, double g[16]) {
    switch (m->kind) {
    case ORC_KERR_BL: contravariant_bl(m, r, theta, g); break;
    case ORC_KERR_KS: contravariant_ks(m, r, theta, g); break;
    default: contravariant_schw(m, r, theta, g); break;
    }
}

void orc_hamiltonian_derivatives(const orc_metric *m, double r, double theta, const double p[4],
                                 double *dh_dr, double *dh_dtheta) {
    switch (m->kind) {
    case ORC_KERR_BL: hamiltonian_derivs_bl(m, r, theta, p, dh_dr, dh_dtheta); break;
    case ORC_KERR_KS: hamiltonian_derivs_ks(m, r, theta, p, dh_dr, dh_dtheta); break;
    default: hamiltonian_derivs_schw(m, r, theta, p, dh_dr, dh_dtheta); break;
    }
}

/* tensor/metric_tensor.rs:52-60 */
double orc_contract(const double g[16], const double p[4]) {
    double result = 0.0;
    for (int mu = 0; mu < 4; mu++)
        for (int nu = 0; nu < 4; nu++) result += g[mu * 4 + nu] * p[mu] * p[nu];
    return result;
}

/* metric/kerr.rs:181-189 */
double orc_time_dilation(const orc_metric *m, double r, double theta) {
    double g[16];
    orc_covariant(m, r, theta, g);
    double g_tt = g[0];
    if (g_tt >= 0.0) return 0.0;
    return sqrt(-g_tt);
}

/* gravitas-wasm/src/lib.rs:97-105 (metric_bl at theta = pi/2) */
double orc_compute_dilation(const orc_metric *m, double r) {
    orc_metric bl = *m;
    if (bl.kind != ORC_SCHWARZSCHILD) bl.kind = ORC_KERR_BL;
    double td = orc_time_dilation(&bl, r, 1.57079632679489661923);
    if (td <= 0.0) return 100.0;
    return 1.0 / td;
}

/* --------------------------------------------------------------- geodesic */

/* geodesic/hamiltonian.rs:13-35 */
orc_state orc_state_derivative(const orc_state *s, const orc_metric *m) {
    double r = s->x[1];
    double theta = s->x[2];
    double g[16];
    orc_contravariant(m, r, theta, g);
    const double *p = s->p;

    double dt = g[0] * p[0] + g[1] * p[1] + g[3] * p[3];
    double dr = g[4] * p[0] + g[5] * p[1] + g[7] * p[3];
    double dth = g[10] * p[2];
    double dph = g[12] * p[0] + g[13] * p[1] + g[15] * p[3];

    double dh_dr, dh_dth;
    orc_hamiltonian_derivatives(m, r, theta, s->p, &dh_dr, &dh_dth);

    orc_state d;
    d.x[0] = dt;
    d.x[1] = dr;
    d.x[2] = dth;
    d.x[3] = dph;
    d.p[0] = 0.0;
    d.p[1] = -dh_dr;
    d.p[2] = -dh_dth;
    d.p[3] = 0.0;
    return d;
}

/* invariants/mod.rs:25-37 */
double orc_hamiltonian(const orc_state *s, const orc_metric *m) {
    double g[16];
    orc_contravariant(m, s->x[1], s->x[2], g);
    const double *p = s->p;
    return 0.5 * (g[0] * p[0] * p[0] + g[5] * p[1] * p[1] + g[10] * p[2] * p[2] +
                  g[15] * p[3] * p[3] + 2.0 * g[3] * p[0] * p[3] + 2.0 * g[1] * p[0] * p[1] +
                  2.0 * g[7] * p[1] * p[3]);
}

/* invariants/renormalization.rs:13-45 */
void orc_renormalize_null(orc_state *s, const orc_metric *m) {
    double r = s->x[1];
    double theta = s->x[2];
    double g[16];
    orc_contravariant(m, r, theta, g);

    double p_t = s->p[0];
    double p_r = s->p[1];
    double p_th = s->p[2];
    double p_ph = s->p[3];

    double a_quad = g[5];
    double b_quad = 2.0 * (g[1] * p_t + g[7] * p_ph);
    double c_quad =
        g[0] * p_t * p_t + g[10] * p_th * p_th + g[15] * p_ph * p_ph + 2.0 * g[3] * p_t * p_ph;

    if (fabs(a_quad) > 1e-12) {
        double discriminant = b_quad * b_quad - 4.0 * a_quad * c_quad;
        if (discriminant >= 0.0) {
            double sqrt_d = sqrt(discriminant);
            double sol1 = (-b_quad + sqrt_d) / (2.0 * a_quad);
            double sol2 = (-b_quad - sqrt_d) / (2.0 * a_quad);
            s->p[1] = (fabs(sol1 - p_r) < fabs(sol2 - p_r)) ? sol1 : sol2;
        }
    }
}

/* invariants/constants_of_motion.rs:23-45 (Carter Q only) */
double orc_carter_constant(const orc_state *s, const orc_metric *m) {
    double p_t = s->p[0];
    double p_th = s->p[2];
    double p_ph = s->p[3];
    double theta = s->x[2];
    double a = m->spin * m->mass;
    double cos_theta = orc_cos(theta);
    double sin_theta = orc_sin(theta);
    double sin2 = sin_theta * sin_theta;
    double energy = -p_t;
    double e2 = energy * energy;
    double lz2 = p_ph * p_ph;
    double lz_term = (sin2 < 1e-12) ? 0.0 : lz2 / sin2;
    return p_th * p_th + cos_theta * cos_theta * (lz_term - a * a * e2);
}

/* geodesic/mod.rs:72-145 (add_scaled .. add_scaled_5) */
static orc_state add_scaled(const orc_state *s, const orc_state *k, double c) {
    orc_state n = *s;
    for (int i = 0; i < 4; i++) {
        n.x[i] += k->x[i] * c;
        n.p[i] += k->p[i] * c;
    }
    return n;
}
static orc_state add_scaled_2(const orc_state *s, const orc_state *k1, double s1,
                              const orc_state *k2, double s2) {
    orc_state n = *s;
    for (int i = 0; i < 4; i++) {
        n.x[i] += k1->x[i] * s1 + k2->x[i] * s2;
        n.p[i] += k1->p[i] * s1 + k2->p[i] * s2;
    }
    return n;
}
static orc_state add_scaled_3(const orc_state *s, const orc_state *k1, double s1,
                              const orc_state *k2, double s2, const orc_state *k3, double s3) {
    orc_state n = *s;
    for (int i = 0; i < 4; i++) {
        n.x[i] += k1->x[i] * s1 + k2->x[i] * s2 + k3->x[i] * s3;
        n.p[i] += k1->p[i] * s1 + k2->p[i] * s2 + k3->p[i] * s3;
    }
    return n;
}
static orc_state add_scaled_4(const orc_state *s, const orc_state *k1, double s1,
                              const orc_state *k2, double s2, const orc_state *k3, double s3,
                              const orc_state *k4, double s4) {
    orc_state n = *s;
    for (int i = 0; i < 4; i++) {
        n.x[i] += k1->x[i] * s1 + k2->x[i] * s2 + k3->x[i] * s3 + k4->x[i] * s4;
        n.p[i] += k1->p[i] * s1 + k2->p[i] * s2 + k3->p[i] * s3 + k4->p[i] * s4;
    }
    return n;
}
static orc_state add_scaled_5(const orc_state *s, const orc_state *k1, double s1,
                              const orc_state *k2, double s2, const orc_state *k3, double s3,
                              const orc_state *k4, double s4, const orc_state *k5, double s5) {
    orc_state n = *s;
    for (int i = 0; i < 4; i++) {
        n.x[i] += k1->x[i] * s1 + k2->x[i] * s2 + k3->x[i] * s3 + k4->x[i] * s4 + k5->x[i] * s5;
        n.p[i] += k1->p[i] * s1 + k2->p[i] * s2 + k3->p[i] * s3 + k4->p[i] * s4 + k5->p[i] * s5;
    }
    return n;
}

/* geodesic/integrator.rs:113-190 ; returns the error estimate */
double orc_rkf45_step(const orc_state *state, const orc_metric *m, double h, orc_state *out) {
    orc_state t;
    orc_state k1 = orc_state_derivative(state, m);
    t = add_scaled(state, &k1, h / 4.0);
    orc_state k2 = orc_state_derivative(&t, m);
    t = add_scaled_2(state, &k1, 3.0 * h / 32.0, &k2, 9.0 * h / 32.0);
    orc_state k3 = orc_state_derivative(&t, m);
    t = add_scaled_3(state, &k1, 1932.0 * h / 2197.0, &k2, -7200.0 * h / 2197.0, &k3,
                     7296.0 * h / 2197.0);
    orc_state k4 = orc_state_derivative(&t, m);
    t = add_scaled_4(state, &k1, 439.0 * h / 216.0, &k2, -8.0 * h, &k3, 3680.0 * h / 513.0, &k4,
                     -845.0 * h / 4104.0);
    orc_state k5 = orc_state_derivative(&t, m);
    t = add_scaled_5(state, &k1, -8.0 * h / 27.0, &k2, 2.0 * h, &k3, -3544.0 * h / 2565.0, &k4,
                     1859.0 * h / 4104.0, &k5, -11.0 * h / 40.0);
    orc_state k6 = orc_state_derivative(&t, m);

    orc_state fs = *state;
    for (int i = 0; i < 4; i++) {
        fs.x[i] += h * (16.0 / 135.0 * k1.x[i] + 6656.0 / 12825.0 * k3.x[i] +
                        28561.0 / 56430.0 * k4.x[i] - 9.0 / 50.0 * k5.x[i] + 2.0 / 55.0 * k6.x[i]);
        fs.p[i] += h * (16.0 / 135.0 * k1.p[i] + 6656.0 / 12825.0 * k3.p[i] +
                        28561.0 / 56430.0 * k4.p[i] - 9.0 / 50.0 * k5.p[i] + 2.0 / 55.0 * k6.p[i]);
    }

    double error = 0.0;
    for (int i = 0; i < 4; i++) {
        double err = h * ((16.0 / 135.0 - 25.0 / 216.0) * k1.x[i] +
                          (6656.0 / 12825.0 - 1408.0 / 2565.0) * k3.x[i] +
                          (28561.0 / 56430.0 - 2197.0 / 4104.0) * k4.x[i] +
                          (-9.0 / 50.0 + 1.0 / 5.0) * k5.x[i] + 2.0 / 55.0 * k6.x[i]);
        error = fmax(error, fabs(err));
    }
    *out = fs;
    return error;
}

/* geodesic/integrator.rs:53-107 ; constants :61-68 ; returns next h */
double orc_adaptive_step(orc_state *state, const orc_metric *m, double h_try, double tolerance,
                         uint64_t *tries) {
    const double safety_factor = 0.9;
    const double min_step = 1e-5;
    const double max_step = 10.0;
    double h = rs_clamp(h_try, -max_step, max_step);

    for (;;) {
        orc_state new_state;
        double error_estimate = orc_rkf45_step(state, m, h, &new_state);
        if (tries) (*tries)++;

        double error_ratio = (error_estimate == 0.0) ? 0.0 : error_estimate / tolerance;

        if (error_ratio <= 1.0) {
            *state = new_state;
            double growth = (error_ratio < 1e-4) ? 5.0 : safety_factor * orc_pow(error_ratio, -0.2);
            double next_h = h * fmin(growth, 5.0);
            return rs_clamp(next_h, -max_step, max_step);
        } else {
            double shrink = safety_factor * orc_pow(error_ratio, -0.25);
            h *= fmax(shrink, 0.1);
            if (fabs(h) < min_step) {
                orc_state forced;
                (void)orc_rkf45_step(state, m, min_step * rs_signum(h), &forced);
                if (tries) (*tries)++;
                *state = forced;
                return min_step * rs_signum(h);
            }
        }
    }
}

/* geodesic/integrator.rs:193-203 */
void orc_step_rk4(orc_state *state, const orc_metric *m, double h) {
    orc_state t;
    orc_state k1 = orc_state_derivative(state, m);
    t = add_scaled(state, &k1, 0.5 * h);
    orc_state k2 = orc_state_derivative(&t, m);
    t = add_scaled(state, &k2, 0.5 * h);
    orc_state k3 = orc_state_derivative(&t, m);
    t = add_scaled(state, &k3, h);
    orc_state k4 = orc_state_derivative(&t, m);
    for (int i = 0; i < 4; i++) {
        state->x[i] += (h / 6.0) * (k1.x[i] + 2.0 * k2.x[i] + 2.0 * k3.x[i] + k4.x[i]);
        state->p[i] += (h / 6.0) * (k1.p[i] + 2.0 * k2.p[i] + 2.0 * k3.p[i] + k4.p[i]);
    }
}

/* geodesic/integrator.rs:209-226 */
void orc_step_symplectic(orc_state *state, const orc_metric *m, double h) {
    orc_state s_mid = *state;
    for (int it = 0; it < 2; it++) {
        orc_state d = orc_state_derivative(&s_mid, m);
        orc_state s_next = *state;
        for (int i = 0; i < 4; i++) {
            s_next.x[i] = state->x[i] + d.x[i] * h;
            s_next.p[i] = state->p[i] + d.p[i] * h;
            s_mid.x[i] = 0.5 * (state->x[i] + s_next.x[i]);
            s_mid.p[i] = 0.5 * (state->p[i] + s_next.p[i]);
        }
    }
    orc_state d_final = orc_state_derivative(&s_mid, m);
    for (int i = 0; i < 4; i++) {
        state->x[i] += d_final.x[i] * h;
        state->p[i] += d_final.p[i] * h;
    }
}

/* geodesic/mod.rs:256-265 */
static int check_termination(const orc_state *s, double horizon, double escape_r) {
    double r = s->x[1];
    if (r < horizon * 1.001) return ORC_TERM_HORIZON;
    if (r > escape_r) return ORC_TERM_ESCAPE;
    return ORC_TERM_NONE;
}

/* geodesic/mod.rs:180-253 */
size_t orc_integrate_path(const orc_state *initial, const orc_metric *m, const orc_options *opt,
                          orc_trajectory *out, orc_state *path, size_t cap) {
    orc_state state = *initial;
    double h = opt->initial_step;
    double horizon = orc_event_horizon(m);
    double max_drift = 0.0;
    uint64_t steps = 0;
    uint64_t tries = 0;
    size_t npath = 0;

    if (path && npath < cap) path[npath++] = state; /* mod.rs:193-197 (pre-renormalisation) */

    orc_renormalize_null(&state, m); /* mod.rs:200 */

    for (uint64_t it = 0; it < opt->max_steps; it++) {
        int term = check_termination(&state, horizon, opt->escape_radius);
        if (term != ORC_TERM_NONE) {
            out->final_state = state;
            out->termination = term;
            out->steps_taken = steps;
            out->max_hamiltonian_drift = max_drift;
            out->rkf_tries = tries;
            return npath;
        }

        switch (opt->method) {
        case ORC_METHOD_RKF45: h = orc_adaptive_step(&state, m, h, opt->tolerance, &tries); break;
        case ORC_METHOD_RK4: orc_step_rk4(&state, m, opt->step_size); break;
        default: orc_step_symplectic(&state, m, opt->step_size); break;
        }

        /* mod.rs:229-231 (Rust would panic on interval 0; we treat 0 as "never") */
        if (opt->renormalize_interval != 0 && steps % opt->renormalize_interval == 0)
            orc_renormalize_null(&state, m);

        double h_val = fabs(orc_hamiltonian(&state, m));
        if (h_val > max_drift) max_drift = h_val;

        steps += 1;
        if (path && npath < cap) path[npath++] = state;
    }

    out->final_state = state;
    out->termination = ORC_TERM_MAXSTEPS;
    out->steps_taken = steps;
    out->max_hamiltonian_drift = max_drift;
    out->rkf_tries = tries;
    return npath;
}

void orc_integrate(const orc_state *initial, const orc_metric *m, const orc_options *opt,
                   orc_trajectory *out) {
    (void)orc_integrate_path(initial, m, opt, out, NULL, 0);
}

/* gravitas-wasm/src/lib.rs:422-464 */
size_t orc_integrate_ray_relativistic(double mass, double spin, const double *initial, size_t n,
                                      uint64_t steps, double tolerance, int use_kerr_schild,
                                      double *out) {
    if (n < 8) { /* lib.rs:429-431: echo the input */
        for (size_t i = 0; i < n; i++) out[i] = initial[i];
        return n;
    }
    orc_state s;
    for (int i = 0; i < 4; i++) {
        s.x[i] = initial[i];
        s.p[i] = initial[4 + i];
    }
    orc_options o; /* lib.rs:444-452 */
    o.method = ORC_METHOD_RKF45;
    o.tolerance = tolerance;
    o.initial_step = 0.01;
    o.max_steps = steps;
    o.escape_radius = 1000.0;
    o.renormalize_interval = 10;
    o.step_size = 0.0;
    orc_metric m = orc_metric_make(use_kerr_schild ? ORC_KERR_KS : ORC_KERR_BL, mass, spin);
    orc_trajectory tr;
    orc_integrate(&s, &m, &o, &tr);
    for (int i = 0; i < 4; i++) {
        out[i] = tr.final_state.x[i];
        out[4 + i] = tr.final_state.p[i];
    }
    return 8;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_integrate_batch(const orc_metric *m, const orc_options *opt, size_t n,
                         const orc_state *in, orc_state *out, uint32_t *steps, uint8_t *term,
                         double *drift, uint32_t *tries, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    long long nn = (long long)n;
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
    for (long long i = 0; i < nn; i++) {
        orc_trajectory tr;
        orc_integrate(&in[i], m, opt, &tr);
        out[i] = tr.final_state;
        if (steps) steps[i] = (uint32_t)tr.steps_taken;
        if (term) term[i] = (uint8_t)tr.termination;
        if (drift) drift[i] = tr.max_hamiltonian_drift;
        if (tries) tries[i] = (uint32_t)tr.rkf_tries;
    }
}

/* ------------------------------------------------------ physics/redshift.rs */

/* physics/redshift.rs:17-23 */
double orc_gravitational_factor(double r, double mass) {
    double rs = 2.0 * mass;
    if (r <= rs) return 0.0;
    return sqrt(1.0 - rs / r);
}

/* physics/redshift.rs:32-35 */
double orc_doppler_factor(double beta, double cos_theta) {
    double gamma = 1.0 / sqrt(fmax(1.0 - beta * beta, 1e-12));
    return 1.0 / (gamma * (1.0 - beta * cos_theta));
}

/* physics/redshift.rs:65-95 */
double orc_kerr_g_factor(double r, double mass, double spin, double lambda) {
    double a = spin * mass;
    double r2 = r * r;
    double a2 = a * a;
    double m = mass;

    double omega = sqrt(m) / (orc_pow(r, 1.5) + a * sqrt(m));

    double sigma = r2;
    double g_tt = -(1.0 - 2.0 * m * r / sigma);
    double g_tphi = -(2.0 * m * r * a) / sigma;
    double g_phiphi = r2 + a2 + 2.0 * m * r * a2 / sigma;

    double ut_denom = -g_tt - 2.0 * omega * g_tphi - omega * omega * g_phiphi;
    if (ut_denom <= 0.0) return 0.0;
    double ut = 1.0 / sqrt(ut_denom);

    double factor = 1.0 - lambda * omega;
    if (fabs(factor) < 1e-30) return 0.0;

    return 1.0 / (ut * factor);
}

/* physics/redshift.rs:126-132 */
double orc_intensity_scaling(double g, int optically_thick) {
    return optically_thick ? rs_powi(g, 4) : rs_powi(g, 3);
}

/* ------------------------------------------------------ physics/spectrum.rs */

/* constants.rs:22,31 ; physics/spectrum.rs:5-7 */
#define ORC_SI_C 299792458.0
#define ORC_SI_KB 1.380649e-23
#define ORC_H 6.62607015e-34
static const double ORC_C1 = 2.0 * ORC_H * ORC_SI_C * ORC_SI_C;
static const double ORC_C2 = ORC_H * ORC_SI_C / ORC_SI_KB;

/* physics/spectrum.rs:12-18 */
double orc_planck_law(double lambda, double temperature) {
    double exponent = ORC_C2 / (lambda * temperature);
    if (exponent > 100.0) return 0.0;
    return (ORC_C1 / rs_powi(lambda, 5)) / (orc_exp(exponent) - 1.0);
}

static double cie_lobe(double l_nm, double mean, double std) {
    double x = (l_nm - mean) / std;
    return orc_exp(-0.5 * x * x);
}

/* physics/spectrum.rs:50-62 */
void orc_cie_1931(double lambda, double xyz[3]) {
    double l_nm = lambda * 1e9;
    double x = 1.056 * cie_lobe(l_nm, 599.0, 37.9) + 0.362 * cie_lobe(l_nm, 442.0, 16.0) -
               0.065 * cie_lobe(l_nm, 501.0, 20.4);
    double y = 0.821 * cie_lobe(l_nm, 568.0, 46.9) + 0.286 * cie_lobe(l_nm, 530.0, 22.1);
    double z = 1.217 * cie_lobe(l_nm, 437.0, 11.8) + 0.681 * cie_lobe(l_nm, 459.0, 26.0);
    xyz[0] = fmax(x, 0.0);
    xyz[1] = fmax(y, 0.0);
    xyz[2] = fmax(z, 0.0);
}

/* physics/spectrum.rs:23-47 */
void orc_integrate_planck_xyz(double temperature, double xyz[3]) {
    xyz[0] = xyz[1] = xyz[2] = 0.0;
    if (temperature < 100.0) return;
    double x = 0.0, y = 0.0, z = 0.0;
    double lambda = 380.0e-9;
    double end = 780.0e-9;
    double step = 2.0e-9;
    while (lambda <= end) {
        double intensity = orc_planck_law(lambda, temperature);
        double c[3];
        orc_cie_1931(lambda, c);
        x += intensity * c[0] * step;
        y += intensity * c[1] * step;
        z += intensity * c[2] * step;
        lambda += step;
    }
    xyz[0] = x;
    xyz[1] = y;
    xyz[2] = z;
}

/* physics/spectrum.rs:65-70 */
void orc_xyz_to_linear_rgb(double x, double y, double z, float rgb[3]) {
    double r = 3.2404542 * x - 1.5371385 * y - 0.4985314 * z;
    double g = -0.9692660 * x + 1.8760108 * y + 0.0415560 * z;
    double b = 0.0556434 * x - 0.2040259 * y + 1.0572252 * z;
    rgb[0] = (float)fmax(r, 0.0);
    rgb[1] = (float)fmax(g, 0.0);
    rgb[2] = (float)fmax(b, 0.0);
}

/* physics/spectrum.rs:76-102 */
void orc_generate_blackbody_lut(size_t width, size_t height, double max_temp, float *out) {
    const double min_g = 0.05;
    const double max_g = 5.0;
    size_t hden = (height - 1) > 1 ? (height - 1) : 1; /* (height - 1).max(1) */
    size_t wden = (width - 1) > 1 ? (width - 1) : 1;
    if (height == 0 || width == 0) return;
    if (height == 1) hden = 1;
    if (width == 1) wden = 1;
    size_t k = 0;
    for (size_t y = 0; y < height; y++) {
        double g = min_g + (max_g - min_g) * ((double)y / (double)hden);
        for (size_t x = 0; x < width; x++) {
            double t = orc_pow((double)x / (double)wden, 2.5) * max_temp;
            double t_eff = t * g;
            double xyz[3];
            orc_integrate_planck_xyz(t_eff, xyz);
            float rgb[3];
            orc_xyz_to_linear_rgb(xyz[0], xyz[1], xyz[2], rgb);
            double g4 = rs_powi(g, 4);
            double scale = 1.0e-14 * g4;
            out[k++] = rgb[0] * (float)scale;
            out[k++] = rgb[1] * (float)scale;
            out[k++] = rgb[2] * (float)scale;
            out[k++] = 1.0f;
        }
    }
}
