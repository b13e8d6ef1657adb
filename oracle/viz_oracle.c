/* viz_oracle.c -- see viz_oracle.h.  TEST INFRASTRUCTURE ONLY.  Plain C, reference operation order. */
#include "viz_oracle.h"
#include "ref_libm.h"

#include <math.h>

#define ORC_PI 3.14159265358979323846
#define ORC_PI_2 1.57079632679489661923

typedef struct {
    double tt, rr, thth, phph, tph;
} cov_bl;

/* Kerr::covariant_bl, metric/kerr.rs:241-264 */
static cov_bl covariant_bl(double r, double theta, double m, double a) {
    const double r2 = r * r, a2 = a * a;
    const double sin_theta = orc_sin(theta), cos_theta = orc_cos(theta);
    const double sin2 = sin_theta * sin_theta, cos2 = cos_theta * cos_theta;
    const double sigma = r2 + a2 * cos2;
    const double delta = r2 - 2.0 * m * r + a2;
    cov_bl g;
    g.tt = -(1.0 - (2.0 * m * r) / sigma);
    g.rr = sigma / delta;
    g.thth = sigma;
    g.phph = (r2 + a2 + (2.0 * m * r * a2 * sin2) / sigma) * sin2;
    g.tph = -(2.0 * m * r * a * sin2) / sigma;
    return g;
}

static double clamp_spin(double s) { return s < -1.0 ? -1.0 : (s > 1.0 ? 1.0 : s); }

/* spacetime/curvature.rs:22-47; x.powi(6) = x^2 * (x^2)^2 (binary exponentiation) */
double orc_kretschner_kerr(double r, double theta, double mass, double spin) {
    const double a = spin * mass;
    const double r2 = r * r, a2 = a * a;
    const double cos_theta = orc_cos(theta);
    const double cos2 = cos_theta * cos_theta, cos4 = cos2 * cos2, cos6 = cos4 * cos2;
    const double r4 = r2 * r2, r6 = r4 * r2;
    const double a4 = a2 * a2, a6 = a4 * a2;
    const double sigma = r2 + a2 * cos2;
    const double s2 = sigma * sigma;
    const double sigma6 = s2 * (s2 * s2);
    if (sigma6 < 1e-30) return INFINITY;
    const double numerator = r6 - 15.0 * r4 * a2 * cos2 + 15.0 * r2 * a4 * cos4 - a6 * cos6;
    return 48.0 * mass * mass * numerator / sigma6;
}

/* spacetime/lightcone.rs:19-48 with the engine's Boyer-Lindquist metric (g_tr = 0 exactly) */
double orc_light_cone_tilt_bl(double r, double theta, double mass, double spin) {
    const cov_bl g = covariant_bl(r, theta, mass, clamp_spin(spin) * mass);
    const double g_tr = 0.0;
    if (fabs(g_tr) < 1e-12) {
        if (g.tt >= 0.0) return ORC_PI_2;
        const double ratio = fmax(-g.tt / g.rr, 0.0);
        return orc_atan(sqrt(ratio));
    } else {
        const double disc = g_tr * g_tr - g.tt * g.rr;
        if (disc < 0.0) return ORC_PI_2;
        const double sq = sqrt(disc);
        const double slope_out = (-g_tr + sq) / g.rr, slope_in = (-g_tr - sq) / g.rr;
        return orc_atan(fabs(slope_out - slope_in) / 2.0);
    }
}

/* Kerr::frame_dragging, metric/kerr.rs:143-152 */
double orc_frame_dragging_omega(double r, double theta, double mass, double spin) {
    const cov_bl g = covariant_bl(r, theta, mass, clamp_spin(spin) * mass);
    if (fabs(g.phph) < 1e-30) return 0.0;
    return -g.tph / g.phph;
}

/* Kerr::ergosphere, metric/kerr.rs:157-167 */
double orc_ergosphere_radius(double theta, double mass, double spin) {
    const double a = clamp_spin(spin) * mass;
    const double c = orc_cos(theta);
    const double disc = mass * mass - a * a * c * c;
    return disc < 0.0 ? mass : mass + sqrt(disc);
}

/* spacetime/embedding.rs:17-23 */
double orc_flamm_height(double r, double mass) {
    const double rs = 2.0 * mass;
    if (r <= rs) return 0.0;
    return 2.0 * sqrt(rs * (r - rs));
}

/* spacetime/embedding.rs:31-46 */
double orc_kerr_embedding_height(double r, double r_ref, size_t n_steps, double mass, double spin) {
    const double a = clamp_spin(spin) * mass;
    const double dr = (r_ref - r) / (double)n_steps;
    double z = 0.0;
    for (size_t i = 0; i < n_steps; ++i) {
        const double r_i = r + ((double)i + 0.5) * dr;
        const cov_bl g = covariant_bl(r_i, ORC_PI_2, mass, a);
        z += sqrt(fabs(g.rr - 1.0)) * dr;
    }
    return z;
}

/* spacetime/embedding.rs:51-65 */
double orc_proper_distance(double r1, double r2, size_t n_steps, double mass, double spin) {
    const double a = clamp_spin(spin) * mass;
    const double r_lo = r1 < r2 ? r1 : r2, r_hi = r1 < r2 ? r2 : r1;
    const double dr = (r_hi - r_lo) / (double)n_steps;
    double dist = 0.0;
    for (size_t i = 0; i < n_steps; ++i) {
        const double r_i = r_lo + ((double)i + 0.5) * dr;
        const cov_bl g = covariant_bl(r_i, ORC_PI_2, mass, a);
        dist += sqrt(fabs(g.rr)) * dr;
    }
    return dist;
}

/* curvature.rs:49-68, lightcone.rs:53-73, frame_drag.rs:15-34 (same grid) */
void orc_scalar_field(int kind, double mass, double spin, double r_min, double r_max,
                      size_t n_radial, size_t n_polar, float *out) {
    for (size_t i = 0; i < n_radial; ++i) {
        const double r = r_min + (r_max - r_min) * (double)i / (double)(n_radial - 1);
        for (size_t j = 0; j < n_polar; ++j) {
            const double theta = 0.1 + (ORC_PI - 0.2) * (double)j / (double)(n_polar - 1);
            double v;
            if (kind == 0) v = orc_kretschner_kerr(r, theta, mass, spin);
            else if (kind == 1) v = orc_light_cone_tilt_bl(r, theta, mass, spin);
            else v = orc_frame_dragging_omega(r, theta, mass, spin);
            float *o = out + 3 * (i * n_polar + j);
            o[0] = (float)r;
            o[1] = (float)theta;
            o[2] = (float)v;
        }
    }
}

/* spacetime/embedding.rs:70-111 */
void orc_embedding_mesh(double mass, double spin_raw, double r_min, double r_max, size_t n_radial,
                        size_t n_angular, float *out) {
    for (size_t i = 0; i < n_radial; ++i) {
        const double t = (double)i / (double)(n_radial - 1);
        const double r = r_min + t * (r_max - r_min);
        const double height = fabs(spin_raw) < 1e-6
                                  ? orc_flamm_height(r, mass)
                                  : orc_kerr_embedding_height(r, r_max, 100, mass, spin_raw);
        for (size_t j = 0; j < n_angular; ++j) {
            const double phi = 2.0 * ORC_PI * (double)j / (double)n_angular;
            float *o = out + 3 * (i * n_angular + j);
            o[0] = (float)(r * orc_cos(phi));
            o[1] = (float)(-height);
            o[2] = (float)(r * orc_sin(phi));
        }
    }
}

/* spacetime/frame_drag.rs:40-69 */
void orc_ergosphere_mesh(double mass, double spin, size_t n_polar, size_t n_azimuthal, float *out) {
    for (size_t i = 0; i < n_polar; ++i) {
        const double theta = ORC_PI * (double)i / (double)(n_polar - 1);
        const double r_ergo = orc_ergosphere_radius(theta, mass, spin);
        for (size_t j = 0; j < n_azimuthal; ++j) {
            const double phi = 2.0 * ORC_PI * (double)j / (double)n_azimuthal;
            float *o = out + 3 * (i * n_azimuthal + j);
            o[0] = (float)(r_ergo * orc_sin(theta) * orc_cos(phi));
            o[1] = (float)(r_ergo * orc_cos(theta));
            o[2] = (float)(r_ergo * orc_sin(theta) * orc_sin(phi));
        }
    }
}
