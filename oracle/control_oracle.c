/*
 * control_oracle.c -- see control_oracle.h.  TEST INFRASTRUCTURE ONLY.
 * Citations: file:line under /root/reference/physics-engine/.
 */
#include "control_oracle.h"
#include "ref_libm.h"

#include <math.h>
#include <string.h>

#define ORC_PI 3.14159265358979323846

static double powi2(double x) { return x * x; }

/* gravitas-core/src/physics/disk.rs:24-36 */
static double specific_energy(double r, double m, double a) {
    double rm = r / m;
    double sqrt_mr = sqrt(m / r);
    double am = a / m;
    double num = 1.0 - 2.0 / rm + am * sqrt_mr;
    double den_sq = 1.0 - 3.0 / rm + 2.0 * am * sqrt_mr;
    if (den_sq <= 0.0) return 1.0;
    return num / sqrt(den_sq);
}

/* disk.rs:44-57 */
static double specific_angular_momentum(double r, double m, double a) {
    double rm = r / m;
    double sqrt_mr = sqrt(m / r);
    double am = a / m;
    double a2r2 = powi2(a / r);
    double num = sqrt(m) * sqrt(r) * (1.0 - 2.0 * am * sqrt_mr + a2r2);
    double den_sq = 1.0 - 3.0 / rm + 2.0 * am * sqrt_mr;
    if (den_sq <= 0.0) return 0.0;
    return num / sqrt(den_sq);
}

/* disk.rs:62-64 */
static double angular_velocity(double r, double m, double a) {
    return sqrt(m) / (orc_pow(r, 1.5) + a * sqrt(m));
}

static double pt_integrand(double rp, double m, double a) { /* disk.rs:122-134 */
    double ep = specific_energy(rp, m, a);
    double lzp = specific_angular_momentum(rp, m, a);
    double omp = angular_velocity(rp, m, a);
    double drp = rp * 1e-5;
    double dlz_dr = (specific_angular_momentum(rp + drp, m, a) -
                     specific_angular_momentum(rp - drp, m, a)) /
                    (2.0 * drp);
    return (ep - omp * lzp) * dlz_dr;
}

/* disk.rs:90-151 */
double orc_page_thorne_flux(double r, double mass, double spin, double m_dot) {
    orc_metric bh = orc_metric_make(ORC_KERR_BL, mass, spin);
    double m = bh.mass;
    double a = bh.spin * bh.mass;
    double r_isco = orc_isco(&bh, 0);
    if (r <= r_isco) return 0.0;

    double e_r = specific_energy(r, m, a);
    double lz_r = specific_angular_momentum(r, m, a);
    double omega_r = angular_velocity(r, m, a);
    double denom = e_r - omega_r * lz_r;
    if (fabs(denom) < 1e-30) return 0.0;

    double dr = r * 1e-5;
    double omega_dr = (angular_velocity(r + dr, m, a) - angular_velocity(r - dr, m, a)) / (2.0 * dr);

    size_t n = 200;
    double h = (r - r_isco) / (double)n;
    if (h <= 0.0) return 0.0;

    double sum = pt_integrand(r_isco, m, a) + pt_integrand(r, m, a);
    for (size_t i = 1; i < n; i++) {
        double rp = r_isco + (double)i * h;
        double weight = (i % 2 == 0) ? 2.0 : 4.0;
        sum += weight * pt_integrand(rp, m, a);
    }
    double integral = sum * h / 3.0;
    double flux = -(omega_dr / (denom * denom)) * integral;
    return fabs(flux) * m_dot;
}

/* disk.rs:160-170 */
double orc_disk_temperature(double r, double mass, double spin, double m_dot) {
    double flux = orc_page_thorne_flux(r, mass, spin, m_dot);
    if (flux <= 0.0) return 0.0;
    double t_scale = 1e7 * orc_pow(m_dot, 0.25);
    return t_scale * orc_pow(flux, 0.25);
}

/* disk.rs:175-201 */
void orc_generate_temperature_lut(double mass, double spin, size_t width, float *out) {
    orc_metric bh = orc_metric_make(ORC_KERR_BL, mass, spin);
    double rin = orc_isco(&bh, 0);
    double rout = 50.0 * bh.mass;
    double max_temp = 0.0;
    size_t den = (width > 1) ? width - 1 : 1;
    double temps[4096];
    if (width > 4096) width = 4096;
    for (size_t i = 0; i < width; i++) {
        double t = (double)i / (double)den;
        double r = rin + t * (rout - rin);
        double temp = orc_disk_temperature(r, mass, spin, 1.0);
        if (temp > max_temp) max_temp = temp;
        temps[i] = temp;
    }
    double norm = (max_temp > 0.0) ? 1.0 / max_temp : 1.0;
    for (size_t i = 0; i < width; i++) out[i] = (float)(temps[i] * norm);
}

/* physics/shadow.rs:191-193 */
double orc_schwarzschild_shadow_radius(double mass) { return 3.0 * sqrt(3.0) * mass; }

/* shadow.rs:39-59 */
static void critical_params(double r, double m, double a, double *xi, double *eta) {
    double r2 = r * r, r3 = r2 * r, a2 = a * a;
    double denom = a * (r - m);
    if (fabs(denom) < 1e-30) {
        *xi = 0.0;
        *eta = 0.0;
        return;
    }
    *xi = -(r3 - 3.0 * m * r2 + a2 * r + a2 * m) / denom;
    double denom2 = a2 * (r - m) * (r - m);
    if (fabs(denom2) < 1e-30) {
        *eta = 0.0;
        return;
    }
    *eta = r3 * (4.0 * m * a2 - r * powi2(r - 3.0 * m)) / denom2;
}

static double beta_sq_of(double r, double m, double a, double sin_obs, double cos_obs) {
    double xi, eta;
    critical_params(r, m, a, &xi, &eta);
    return eta + a * a * cos_obs * cos_obs - xi * xi * cos_obs * cos_obs / (sin_obs * sin_obs);
}

/* shadow.rs:81-183 */
size_t orc_bardeen_shadow(double mass, double spin, double theta_obs, size_t n_points,
                          double *out) {
    orc_metric bh = orc_metric_make(ORC_KERR_BL, mass, spin);
    double m = bh.mass;
    double a = bh.spin * bh.mass;
    double sin_obs = orc_sin(theta_obs), cos_obs = orc_cos(theta_obs);
    size_t k = 0;

    if (fabs(a) < 1e-10) {
        double radius = orc_schwarzschild_shadow_radius(m);
        for (size_t i = 0; i < n_points; i++) {
            double phi = 2.0 * ORC_PI * (double)i / (double)n_points;
            out[2 * k] = radius * orc_cos(phi);
            out[2 * k + 1] = radius * orc_sin(phi);
            k++;
        }
        return k;
    }
    if (fabs(sin_obs) < 1e-10) {
        double r_ph = orc_photon_sphere(&bh);
        double xi, eta;
        critical_params(r_ph, m, a, &xi, &eta);
        double radius = sqrt(fmax(eta + a * a, 0.0));
        for (size_t i = 0; i < 2 * n_points; i++) {
            double phi = 2.0 * ORC_PI * (double)i / (2.0 * (double)n_points);
            out[2 * k] = radius * orc_cos(phi);
            out[2 * k + 1] = radius * orc_sin(phi);
            k++;
        }
        return k;
    }

    double a_star = a / m;
    double r_ph_pro = 2.0 * m * (1.0 + orc_cos((2.0 / 3.0) * orc_acos(-fabs(a_star))));
    double r_ph_retro = 2.0 * m * (1.0 + orc_cos((2.0 / 3.0) * orc_acos(fabs(a_star))));
    double r_min = r_ph_pro, r_max = r_ph_retro;
    int steps = 1000;
    for (int i = 0; i <= steps; i++) {
        double t = (double)i / (double)steps;
        double r = r_ph_pro + t * (r_ph_retro - r_ph_pro);
        if (beta_sq_of(r, m, a, sin_obs, cos_obs) >= 0.0) {
            r_min = r;
            break;
        }
    }
    for (int i = steps; i >= 0; i--) {
        double t = (double)i / (double)steps;
        double r = r_ph_pro + t * (r_ph_retro - r_ph_pro);
        if (beta_sq_of(r, m, a, sin_obs, cos_obs) >= 0.0) {
            r_max = r;
            break;
        }
    }
    size_t den = (n_points > 1) ? n_points - 1 : 1;
    for (size_t i = 0; i < n_points; i++) {
        double phase = ORC_PI * (double)i / (double)den;
        double t = 0.5 - 0.5 * orc_cos(phase);
        double r = r_min + t * (r_max - r_min);
        double xi, eta;
        critical_params(r, m, a, &xi, &eta);
        double alpha = a * sin_obs - xi / sin_obs;
        double beta = sqrt(fmax(beta_sq_of(r, m, a, sin_obs, cos_obs), 0.0));
        out[2 * k] = alpha;
        out[2 * k + 1] = -beta;
        k++;
    }
    for (size_t ii = n_points; ii-- > 0;) {
        double phase = ORC_PI * (double)ii / (double)den;
        double t = 0.5 - 0.5 * orc_cos(phase);
        double r = r_min + t * (r_max - r_min);
        double xi, eta;
        critical_params(r, m, a, &xi, &eta);
        double alpha = a * sin_obs - xi / sin_obs;
        double beta = sqrt(fmax(beta_sq_of(r, m, a, sin_obs, cos_obs), 0.0));
        out[2 * k] = alpha;
        out[2 * k + 1] = beta;
        k++;
    }
    return k;
}

/* gravitas-wasm/src/camera.rs:26-39 */
void orc_sab_engine_init(orc_sab_engine *e, double mass, double spin) {
    memset(e, 0, sizeof *e);
    e->mass = mass;
    e->spin = spin;
    e->camera.position[2] = 20.0;
    e->camera.orientation[1] = 1.0; /* DQuat::from_xyzw(0, 1, 0, 0) */
    e->last_good = e->camera;
}

/* glam 0.24.2 DQuat::from_rotation_y + DQuat::mul_vec3 (scalar path) */
static void rotate_y(double angle, double v[3]) {
    double s = orc_sin(angle * 0.5), c = orc_cos(angle * 0.5);
    double b[3] = {0.0, s, 0.0};
    double w = c;
    double b2 = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
    double dot = v[0] * b[0] + v[1] * b[1] + v[2] * b[2];
    double cr[3] = {b[1] * v[2] - b[2] * v[1], b[2] * v[0] - b[0] * v[2], b[0] * v[1] - b[1] * v[0]};
    double k1 = w * w - b2, k2 = dot * 2.0, k3 = w * 2.0;
    for (int i = 0; i < 3; i++) v[i] = v[i] * k1 + b[i] * k2 + cr[i] * k3;
}

/* camera.rs:42-70 */
void orc_camera_update(orc_camera_state *s, double mouse_dx, double mouse_dy, double zoom_delta,
                       double dt) {
    (void)mouse_dy;
    if (dt <= 0.0) return;
    double friction = orc_exp(-5.0 * dt);
    for (int i = 0; i < 3; i++) s->velocity[i] *= friction;
    for (int i = 0; i < 3; i++) s->position[i] += s->velocity[i] * dt;
    double sensitivity = 2.0;
    double yaw = -mouse_dx * sensitivity * dt;
    rotate_y(yaw, s->position);
    if (s->auto_spin) {
        double spin_rate = 0.15;
        rotate_y(spin_rate * dt, s->position);
    }
    double zoom_factor = 1.0 + zoom_delta * dt;
    for (int i = 0; i < 3; i++) s->position[i] *= zoom_factor;
}

static int cam_valid(const orc_camera_state *s) { /* camera.rs:36-38 */
    for (int i = 0; i < 3; i++)
        if (!isfinite(s->position[i]) || !isfinite(s->velocity[i])) return 0;
    for (int i = 0; i < 4; i++)
        if (!isfinite(s->orientation[i])) return 0;
    return 1;
}

/* gravitas-wasm/src/lib.rs:308-409 (offsets lib.rs:36-40, in f32 elements) */
void orc_tick_sab(orc_sab_engine *e, double dt_override) {
    enum { CONTROL = 0, CAMERA = 64, PHYSICS = 128, TELEMETRY = 256 };
    float *sab = e->sab;
    double mouse_dx = sab[CONTROL + 1];
    double mouse_dy = sab[CONTROL + 2];
    double zoom_delta = sab[CONTROL + 3];
    double dt = (dt_override > 0.0) ? dt_override : (double)sab[CONTROL + 4];
    sab[CONTROL + 1] = 0.0f;
    sab[CONTROL + 2] = 0.0f;
    sab[CONTROL + 3] = 0.0f;

    orc_camera_update(&e->camera, mouse_dx, mouse_dy, zoom_delta, dt);
    if (!cam_valid(&e->camera))
        e->camera = e->last_good;
    else
        e->last_good = e->camera;

    for (int i = 0; i < 3; i++) {
        sab[CAMERA + i] = (float)e->camera.position[i];
        sab[CAMERA + 4 + i] = (float)e->camera.velocity[i];
    }
    for (int i = 0; i < 4; i++) sab[CAMERA + 8 + i] = (float)e->camera.orientation[i];

    orc_metric bl = orc_metric_make(ORC_KERR_BL, e->mass, e->spin);
    sab[PHYSICS] = (float)orc_event_horizon(&bl);
    sab[PHYSICS + 1] = (float)orc_isco(&bl, 0);
    sab[PHYSICS + 2] = (float)e->mass;
    sab[PHYSICS + 3] = (float)e->spin;

    const double *p = e->camera.position;
    double r_cam = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    if (r_cam > 0.0) {
        double cos_theta = p[1] / r_cam;
        double theta_obs = orc_acos(cos_theta);
        double curve[2 * 2 * 32];
        size_t n = orc_bardeen_shadow(e->mass, e->spin, theta_obs, 32, curve);
        for (int i = 0; i < 128; i++) sab[PHYSICS + 16 + i] = 0.0f; /* overruns TELEMETRY: F10 */
        size_t actual = n < 64 ? n : 64;
        sab[PHYSICS + 15] = (float)actual;
        for (size_t i = 0; i < actual; i++) {
            sab[PHYSICS + 16 + i * 2] = (float)curve[2 * i];
            sab[PHYSICS + 16 + i * 2 + 1] = (float)curve[2 * i + 1];
        }
        double min_a = 0.0, max_a = 0.0;
        if (n > 0) {
            min_a = max_a = curve[0];
            for (size_t i = 0; i < n; i++) {
                if (curve[2 * i] < min_a) min_a = curve[2 * i];
                if (curve[2 * i] > max_a) max_a = curve[2 * i];
            }
        }
        sab[PHYSICS + 4] = (float)min_a;
        sab[PHYSICS + 5] = (float)max_a;
    }
    sab[TELEMETRY] += 1.0f;
}
