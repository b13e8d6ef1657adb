/*
 * wgsl_f64_twin.c -- TEST INFRASTRUCTURE (an analysis instrument, not an oracle of its own and
 * never part of the product): the f32 compute march of src/shaders/compute.wgsl.ts:147-258,
 * statement by statement as shader_oracle.c:orc_wgsl_pixel restates it, evaluated in IEEE
 * double with the platform libm.  Same discrete algorithm (implicit midpoint, two sweeps,
 * h = clamp((r - r+) 0.15, 0.05, 1), the same exits), no f32 rounding: where two f32 forms of
 * the march (shader operation order vs the FAST / packed contracts) disagree about a ray --
 * how long it lingers on the unstable photon orbit, whether it ends lit or dark -- this says
 * which of them sits closer to the march the shader's formulas define.
 * Uniforms are the f32 values of orc_wgsl_params promoted to double.
 *
 * exit classes: 0 horizon (r < 1.001 r+), 1 escape (r > 100), 2 step budget, 3 opaque (alpha > 0.99)
 */
#include <math.h>
#include <string.h>

#include "shader_oracle.h"

typedef struct { double x[4], p[4]; } ray64;

static double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* compute.wgsl.ts:42-120 */
static void derivs64(const ray64 *s, double M, double spin, double dx[4], double dp[4]) {
    double a = spin * M;
    double r = s->x[1], theta = s->x[2];
    double r2 = r * r, a2 = a * a;
    double sint = sin(theta), cost = cos(theta);
    double sin2 = fmax(sint * sint, 1e-12);
    double cos2 = 1.0 - sin2;
    double sigma = r2 + a2 * cos2;
    double sigma2 = sigma * sigma;
    double delta = r2 - 2.0 * M * r + a2;
    double g_tt = -(1.0 + 2.0 * M * r / sigma);
    double g_tr = 2.0 * M * r / sigma;
    double g_rr = delta / sigma;
    double g_thth = 1.0 / sigma;
    double g_phph = 1.0 / (sigma * sin2);
    double g_rph = a / sigma;
    const double *p = s->p;
    dx[0] = g_tt * p[0] + g_tr * p[1];
    dx[1] = g_tr * p[0] + g_rr * p[1] + g_rph * p[3];
    dx[2] = g_thth * p[2];
    dx[3] = g_rph * p[1] + g_phph * p[3];
    double dsigma_dr = 2.0 * r;
    double dsigma_dth = -2.0 * a2 * sint * cost;
    double ddelta_dr = 2.0 * r - 2.0 * M;
    double dg_tt_dr = -(2.0 * M * (sigma - r * dsigma_dr)) / sigma2;
    double dg_tt_dth = (2.0 * M * r * dsigma_dth) / sigma2;
    double dg_tr_dr = -dg_tt_dr;
    double dg_tr_dth = -dg_tt_dth;
    double dg_rr_dr = (ddelta_dr * sigma - delta * dsigma_dr) / sigma2;
    double dg_rr_dth = -(delta * dsigma_dth) / sigma2;
    double dg_thth_dr = -dsigma_dr / sigma2;
    double dg_thth_dth = -dsigma_dth / sigma2;
    double dg_phph_dr = -dsigma_dr / (sigma2 * sin2);
    double dg_phph_dth = -(dsigma_dth * sin2 + sigma * sin(2.0 * theta)) / (sigma2 * sin2 * sin2);
    double dg_rph_dr = -(a * dsigma_dr) / sigma2;
    double dg_rph_dth = -(a * dsigma_dth) / sigma2;
    double dh_dr = 0.5 * (dg_tt_dr * p[0] * p[0] + dg_rr_dr * p[1] * p[1] + dg_thth_dr * p[2] * p[2] +
                          dg_phph_dr * p[3] * p[3] + 2.0 * dg_tr_dr * p[0] * p[1] + 2.0 * dg_rph_dr * p[1] * p[3]);
    double dh_dth = 0.5 * (dg_tt_dth * p[0] * p[0] + dg_rr_dth * p[1] * p[1] + dg_thth_dth * p[2] * p[2] +
                           dg_phph_dth * p[3] * p[3] + 2.0 * dg_tr_dth * p[0] * p[1] + 2.0 * dg_rph_dth * p[1] * p[3]);
    dp[0] = 0.0;
    dp[1] = -dh_dr;
    dp[2] = -dh_dth;
    dp[3] = 0.0;
}

/* compute.wgsl.ts:122-133 */
static ray64 symplectic64(const ray64 *s, double h, double M, double spin) {
    ray64 mid = *s;
    double dx[4], dp[4];
    for (int it = 0; it < 2; it++) {
        derivs64(&mid, M, spin, dx, dp);
        for (int k = 0; k < 4; k++) {
            double nx = s->x[k] + dx[k] * h, np = s->p[k] + dp[k] * h;
            mid.x[k] = (s->x[k] + nx) * 0.5;
            mid.p[k] = (s->p[k] + np) * 0.5;
        }
    }
    derivs64(&mid, M, spin, dx, dp);
    ray64 out;
    for (int k = 0; k < 4; k++) {
        out.x[k] = s->x[k] + dx[k] * h;
        out.p[k] = s->p[k] + dp[k] * h;
    }
    return out;
}

static void m4v4d(const float *m, const double v[4], double out[4]) {
    for (int r = 0; r < 4; r++)
        out[r] = (double)m[0 + r] * v[0] + (double)m[4 + r] * v[1] + (double)m[8 + r] * v[2] + (double)m[12 + r] * v[3];
}
static void norm3d(double v[3]) {
    double l = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] /= l; v[1] /= l; v[2] /= l;
}

/* compute.wgsl.ts:147-258 (stars are not drawn: the comparisons this serves run with stars off).
 * out: rgb[3]; exit_class; min_r: the smallest r the ray reached; returns the steps taken. */
static uint32_t pixel_f64(const orc_wgsl_params *P, uint32_t ix, uint32_t iy, double rgb[3],
                          int32_t *exit_class, double *min_r, double *axis_margin);
uint32_t orc_wgsl_pixel_f64(const orc_wgsl_params *P, uint32_t ix, uint32_t iy, double rgb[3],
                            int32_t *exit_class, double *min_r) {
    return pixel_f64(P, ix, iy, rgb, exit_class, min_r, NULL);
}
/* axis_margin: min over the visited states of min(theta, pi - theta): how close the ray came to the
 * polar axis, where the shader's spherical coordinates break down; negative = it stepped across */
static uint32_t pixel_f64(const orc_wgsl_params *P, uint32_t ix, uint32_t iy, double rgb[3],
                          int32_t *exit_class, double *min_r, double *axis_margin) {
    const double PI = (double)3.14159265f; /* the shader's literal, as f32 holds it */
    double fw = (double)P->width, fh = (double)P->height;
    double jx = (double)P->jitter[0] / fw, jy = (double)P->jitter[1] / fh;
    double uvx = (double)ix / fw, uvy = (double)iy / fh;
    double ndcx = (uvx + jx) * 2.0 - 1.0, ndcy = (uvy + jy) * 2.0 - 1.0;
    double clip[4] = {ndcx, -ndcy, 1.0, 1.0}, vt[4];
    m4v4d(P->inv_proj, clip, vt);
    double vd[4] = {vt[0] / vt[3], vt[1] / vt[3], vt[2] / vt[3], 0.0};
    norm3d(vd);
    double wd[4];
    m4v4d(P->inv_view, vd, wd);
    norm3d(wd);
    double cam[3] = {(double)P->position[0], (double)P->position[1], (double)P->position[2]};
    double r0 = sqrt(cam[0] * cam[0] + cam[1] * cam[1] + cam[2] * cam[2]);
    double theta0 = acos(clampd(cam[1] / r0, -1.0, 1.0));
    double phi0 = atan2(cam[2], cam[0]);
    double st = sin(theta0), ct = cos(theta0), sp = sin(phi0), cp = cos(phi0);
    double e_r[3] = {st * cp, ct, st * sp}, e_th[3] = {ct * cp, -st, ct * sp}, e_ph[3] = {-sp, 0.0, cp};
    double pr_far = wd[0] * e_r[0] + wd[1] * e_r[1] + wd[2] * e_r[2];
    double pth_far = (wd[0] * e_th[0] + wd[1] * e_th[1] + wd[2] * e_th[2]) / r0;
    double safe_st = fmax(st, (double)1e-4f);
    double pph_far = (wd[0] * e_ph[0] + wd[1] * e_ph[1] + wd[2] * e_ph[2]) / (r0 * safe_st);
    ray64 s;
    s.x[0] = 0.0; s.x[1] = r0; s.x[2] = theta0; s.x[3] = phi0;
    s.p[0] = -1.0; s.p[1] = pr_far; s.p[2] = pth_far * r0 * r0; s.p[3] = pph_far * r0 * r0 * st * st;

    double M = (double)P->mass, spin = (double)P->spin;
    double a = spin * M;
    double disc = M * M - a * a;
    double rh = disc < 0.0 ? M : M + sqrt(disc); /* compute.wgsl.ts:28-32 */
    double absS = fabs(clampd(a / M, (double)-0.999f, (double)0.999f)); /* :34-40 */
    double z1 = 1.0 + pow(1.0 - absS * absS, 1.0 / 3.0) * (pow(1.0 + absS, 1.0 / 3.0) + pow(1.0 - absS, 1.0 / 3.0));
    double z2 = sqrt(3.0 * absS * absS + z1 * z1);
    double isco = M * (3.0 + z2 - sqrt((3.0 - z1) * (3.0 + z1 + 2.0 * z2)));

    double color[3] = {0.0, 0.0, 0.0}, alpha = 0.0;
    uint32_t steps = 0;
    int32_t cls = 2;
    double rmin = r0, smin = 4.0;
    for (int i = 0; i < P->max_steps; i++) {
        double r = s.x[1];
        if (r < rmin) rmin = r;
        { double mg = fmin(s.x[2], 3.14159265358979323846 - s.x[2]); if (mg < smin) smin = mg; }
        if (r < rh * (double)1.001f) { cls = 0; break; }
        if (r > 100.0) { cls = 1; break; }
        double prev_theta = s.x[2];
        double h = clampd((r - rh) * (double)0.15f, (double)0.05f, 1.0);
        s = symplectic64(&s, h, M, spin);
        steps++;
        double curr_theta = s.x[2];
        if ((prev_theta - PI * 0.5) * (curr_theta - PI * 0.5) <= 0.0 && r > isco && r < 30.0) {
            double Omega = 1.0 / (pow(r, 1.5) + a);
            double u_t = 1.0 / sqrt(fmax(1.0 - 2.0 * M / r - Omega * Omega * (r * r + a * a), (double)1e-4f));
            double u_phi = Omega * u_t;
            double g_factor = -s.p[0] / fmax(-(u_t * s.p[0] + u_phi * s.p[3]), (double)1e-4f);
            double artistic_T = (1.0 / pow(fmax(r / isco, 1.0), 0.75)) * g_factor;
            const double base[3] = {1.0, 0.5, (double)0.1f}, blue[3] = {0.5, (double)0.7f, 1.0}, red[3] = {1.0, (double)0.2f, 0.0};
            double bs = fmax(g_factor - 1.0, 0.0), rs = fmax(1.0 - g_factor, 0.0) * 0.5;
            double target_opacity = (double)0.6f * artistic_T;
            double g4 = pow(g_factor, 4.0);
            double mri_shear = pow(r, -1.5);
            double mri_sat = 1.0 + (double)0.0001f * sin(r * 100.0 * mri_shear);
            for (int c = 0; c < 3; c++) {
                double target = (base[c] + blue[c] * bs - red[c] * rs) * artistic_T * 4.0;
                double I_em = target * target_opacity / fmax(g4, (double)1e-5f);
                double I_obs = g4 * (I_em * mri_sat);
                color[c] += I_obs * (1.0 - alpha);
            }
            alpha += target_opacity * mri_sat;
        }
        if (alpha > (double)0.99f) { cls = 3; break; }
    }
    memcpy(rgb, color, sizeof color);
    if (exit_class) *exit_class = cls;
    if (min_r) *min_r = rmin;
    if (axis_margin) *axis_margin = smin;
    return steps;
}

/* n pixels (xy[2 k], xy[2 k + 1]) -> rgb[3 k ..], steps[k], cls[k], min_r[k], axis_margin[k] (may be NULL) */
void orc_wgsl_pixels_f64(const orc_wgsl_params *p, size_t n, const uint32_t *xy, double *rgb, uint32_t *steps,
                         int32_t *cls, double *min_r, double *axis_margin, int nthreads) {
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads) if (nthreads > 1)
    for (long long k = 0; k < (long long)n; k++)
        steps[k] = pixel_f64(p, xy[2 * k], xy[2 * k + 1], rgb + 3 * k, cls + k, min_r + k,
                             axis_margin ? axis_margin + k : NULL);
}
