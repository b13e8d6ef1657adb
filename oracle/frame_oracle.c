/*
 * frame_oracle.c -- CPU statement of the f64 frame path.  TEST INFRASTRUCTURE
 * ONLY; see frame_oracle.h for what is restated vs defined ("parity unpinned").
 */
#include "frame_oracle.h"
#include "control_oracle.h"
#include "ref_libm.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PI_2 1.57079632679489661923

/* ------------------------------------------------------------ camera setup */

static void v3_normalize(double v[3]) {
    double len = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] /= len;
    v[1] /= len;
    v[2] /= len;
}

/* gl-matrix lookAt + perspective(fovy, aspect, 0.1, 1000) and their inverses,
 * as built in src/components/canvas/WebGPUCanvas.tsx:143-157 (closed-form
 * inverses instead of the general 4x4 inversion). */
void orc_camera_look_at(const double eye[3], const double target[3], const double up[3],
                        double fovy_rad, double aspect, orc_camera *cam) {
    double z[3] = {eye[0] - target[0], eye[1] - target[1], eye[2] - target[2]};
    v3_normalize(z);
    double x[3] = {up[1] * z[2] - up[2] * z[1], up[2] * z[0] - up[0] * z[2],
                   up[0] * z[1] - up[1] * z[0]};
    v3_normalize(x);
    double y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2],
                   z[0] * x[1] - z[1] * x[0]};
    memset(cam, 0, sizeof(*cam));
    for (int k = 0; k < 3; k++) {
        cam->position[k] = eye[k];
        cam->inv_view[0 + k] = x[k];
        cam->inv_view[4 + k] = y[k];
        cam->inv_view[8 + k] = z[k];
        cam->inv_view[12 + k] = eye[k];
    }
    cam->inv_view[15] = 1.0;

    const double near = 0.1, far = 1000.0;
    double f = orc_cos(fovy_rad / 2.0) / orc_sin(fovy_rad / 2.0); /* 1 / tan, specified */
    double a = f / aspect, b = f;
    double c = (far + near) / (near - far);
    double d = 2.0 * far * near / (near - far);
    cam->inv_proj[0] = 1.0 / a;
    cam->inv_proj[5] = 1.0 / b;
    cam->inv_proj[11] = 1.0 / d;
    cam->inv_proj[14] = -1.0;
    cam->inv_proj[15] = c / d;
    cam->pixel_offset[0] = 0.5;
    cam->pixel_offset[1] = 0.5;
}

/* column-major mat4 * vec4 */
static void m4_mul_v4(const double m[16], const double v[4], double out[4]) {
    for (int r = 0; r < 4; r++)
        out[r] = m[0 + r] * v[0] + m[4 + r] * v[1] + m[8 + r] * v[2] + m[12 + r] * v[3];
}

/* src/shaders/compute.wgsl.ts:159-187 in f64 */
void orc_pixel_state(const orc_camera *cam, uint32_t width, uint32_t height, uint32_t i,
                     uint32_t j, orc_state *out) {
    double ux = ((double)i + cam->pixel_offset[0]) / (double)width;
    double uy = ((double)j + cam->pixel_offset[1]) / (double)height;
    double ndc_x = ux * 2.0 - 1.0;
    double ndc_y = uy * 2.0 - 1.0;

    double clip[4] = {ndc_x, -ndc_y, 1.0, 1.0};
    double vt[4];
    m4_mul_v4(cam->inv_proj, clip, vt);
    double vd[4] = {vt[0] / vt[3], vt[1] / vt[3], vt[2] / vt[3], 0.0};
    v3_normalize(vd);
    double wd4[4];
    m4_mul_v4(cam->inv_view, vd, wd4);
    double wd[3] = {wd4[0], wd4[1], wd4[2]};
    v3_normalize(wd);

    const double *cp3 = cam->position;
    double r0 = sqrt(cp3[0] * cp3[0] + cp3[1] * cp3[1] + cp3[2] * cp3[2]);
    double cy = cp3[1] / r0;
    if (cy < -1.0) cy = -1.0;
    if (cy > 1.0) cy = 1.0;
    double theta0 = orc_acos(cy);
    double phi0 = orc_atan2(cp3[2], cp3[0]);

    double st = orc_sin(theta0), ct = orc_cos(theta0);
    double sp = orc_sin(phi0), cp = orc_cos(phi0);

    double pr_far = wd[0] * (st * cp) + wd[1] * ct + wd[2] * (st * sp);
    double pth_far = (wd[0] * (ct * cp) + wd[1] * (-st) + wd[2] * (ct * sp)) / r0;
    double safe_st = fmax(st, 1e-4);
    double pph_far = (wd[0] * (-sp) + wd[1] * 0.0 + wd[2] * cp) / (r0 * safe_st);

    out->x[0] = 0.0;
    out->x[1] = r0;
    out->x[2] = theta0;
    out->x[3] = phi0;
    out->p[0] = -1.0;
    out->p[1] = pr_far;
    out->p[2] = pth_far * r0 * r0;
    out->p[3] = pph_far * r0 * r0 * st * st;
}

/* ----------------------------------------------------------------- shading */

/* src/shaders/blackhole/chunks/disk.ts:100-102 */
double orc_disk_temp_profile(double r, double disk_inner) {
    double isco_r = disk_inner / r;
    if (isco_r < 0.0) isco_r = 0.0;
    if (isco_r > 1.0) isco_r = 1.0;
    double nt_factor = fmax(0.0, 1.0 - sqrt(isco_r));
    return orc_pow(isco_r, 0.75) * orc_pow(nt_factor, 0.25);
}

/* physics/disk.rs:175-201 stores entry i at r = rin + (i/(width-1))(rout-rin); a LINEAR,
 * CLAMP_TO_EDGE fetch interpolates between the two entries around the continuous index */
double orc_disk_lut_profile(const float *lut, uint32_t width, double r, double rin, double rout) {
    double last = (double)(width - 1u);
    double x = (r - rin) / (rout - rin) * last;
    if (!(x > 0.0)) x = 0.0;
    if (x > last) x = last;
    uint32_t i0 = (uint32_t)x;
    uint32_t i1 = (i0 + 1u < width) ? i0 + 1u : i0;
    double t0 = lut[i0], t1 = lut[i1];
    return t0 + (t1 - t0) * (x - (double)i0);
}

/* inverse of the LUT axes of physics/spectrum.rs:82,85:
 *   g = 0.05 + 4.95 * y/(H-1) ; T = (x/(W-1))^2.5 * Tmax */
void orc_lut_sample(const float *lut, uint32_t w, uint32_t h, double max_temp, double temp,
                    double g, double rgb[3]) {
    double u = orc_pow(fmax(temp, 0.0) / max_temp, 1.0 / 2.5);
    double fx = u * (double)(w > 1 ? w - 1 : 1);
    double fy = (g - 0.05) / (5.0 - 0.05) * (double)(h > 1 ? h - 1 : 1);
    if (!(fx > 0.0)) fx = 0.0;
    if (!(fy > 0.0)) fy = 0.0;
    if (fx > (double)(w - 1)) fx = (double)(w - 1);
    if (fy > (double)(h - 1)) fy = (double)(h - 1);
    uint32_t x0 = (uint32_t)fx, y0 = (uint32_t)fy;
    uint32_t x1 = (x0 + 1 < w) ? x0 + 1 : x0;
    uint32_t y1 = (y0 + 1 < h) ? y0 + 1 : y0;
    double tx = fx - (double)x0, ty = fy - (double)y0;
    for (int c = 0; c < 3; c++) {
        double t00 = lut[((size_t)y0 * w + x0) * 4 + c];
        double t10 = lut[((size_t)y0 * w + x1) * 4 + c];
        double t01 = lut[((size_t)y1 * w + x0) * 4 + c];
        double t11 = lut[((size_t)y1 * w + x1) * 4 + c];
        double top = t00 + (t10 - t00) * tx;
        double bot = t01 + (t11 - t01) * tx;
        rgb[c] = top + (bot - top) * ty;
    }
}

/* ------------------------------------------------------------ pixel tracer */

static int check_term(const orc_state *s, double horizon, double escape_r) {
    double r = s->x[1];
    if (r < horizon * 1.001) return ORC_TERM_HORIZON;
    if (r > escape_r) return ORC_TERM_ESCAPE;
    return ORC_TERM_NONE;
}

/* integrate() (gravitas-core/src/geodesic/mod.rs:180-253) with the disk-plane
 * crossing test inserted after each completed step. */
int orc_trace_pixel(const orc_camera *cam, const orc_frame_params *fp, const float *lut,
                    uint32_t i, uint32_t j, orc_state *final_state, uint32_t *steps_out,
                    uint32_t *tries_out, double *drift_out, uint32_t *ncross_out,
                    double *rcross_out, float rgba[4]) {
    orc_metric m = orc_metric_make(fp->metric_kind, fp->mass, fp->spin);
    orc_metric mk = orc_metric_make(ORC_KERR_BL, fp->mass, fp->spin);
    const orc_options *opt = &fp->opt;
    orc_state state;
    orc_pixel_state(cam, fp->width, fp->height, i, j, &state);

    double disk_inner = fp->disk_inner > 0.0 ? fp->disk_inner : orc_isco(&mk, 0);
    double horizon = orc_event_horizon(&m);
    double h = opt->initial_step;
    double max_drift = 0.0;
    uint64_t steps = 0, tries = 0;
    uint32_t ncross = 0;
    double alpha = 0.0;
    double color[3] = {0.0, 0.0, 0.0};
    int term = ORC_TERM_MAXSTEPS;

    orc_renormalize_null(&state, &m);

    for (uint64_t it = 0; it < opt->max_steps; it++) {
        int t = check_term(&state, horizon, opt->escape_radius);
        if (t != ORC_TERM_NONE) {
            term = t;
            break;
        }
        double r_prev = state.x[1];
        double th_prev = state.x[2];

        switch (opt->method) {
        case ORC_METHOD_RKF45:
            h = orc_adaptive_step(&state, &m, h, opt->tolerance, &tries);
            break;
        case ORC_METHOD_RK4: orc_step_rk4(&state, &m, opt->step_size); break;
        default: orc_step_symplectic(&state, &m, opt->step_size); break;
        }
        if (opt->renormalize_interval != 0 && steps % opt->renormalize_interval == 0)
            orc_renormalize_null(&state, &m);
        double h_val = fabs(orc_hamiltonian(&state, &m));
        if (h_val > max_drift) max_drift = h_val;
        steps += 1;

        if (fp->shading) {
            double th_new = state.x[2];
            if ((th_prev - ORC_PI_2) * (th_new - ORC_PI_2) <= 0.0) {
                double dth = th_new - th_prev;
                double f = (dth == 0.0) ? 0.0 : (ORC_PI_2 - th_prev) / dth;
                double r_c = r_prev + f * (state.x[1] - r_prev);
                if (r_c > disk_inner && r_c < fp->disk_outer) {
                    double lambda = state.p[3] / (-state.p[0]);
                    double g = orc_kerr_g_factor(r_c, fp->mass, m.spin, lambda);
                    double prof;
                    if (fp->disk_profile == 1 && fp->disk_lut)
                        prof = orc_disk_lut_profile(fp->disk_lut, 512u, r_c, orc_isco(&mk, 0), 50.0 * mk.mass);
                    else
                        prof = orc_disk_temp_profile(r_c, disk_inner);
                    double temp = fp->disk_temp * prof;
                    double rgb[3] = {0.0, 0.0, 0.0};
                    if (lut)
                        orc_lut_sample(lut, fp->lut_width, fp->lut_height, fp->lut_max_temp, temp,
                                       g, rgb);
                    for (int c = 0; c < 3; c++)
                        color[c] += rgb[c] * fp->exposure * (1.0 - alpha);
                    alpha += fp->disk_opacity;
                    if (rcross_out && ncross < 2) rcross_out[ncross] = r_c;
                    ncross++;
                    if (alpha > 0.99) {
                        term = ORC_TERM_DISK;
                        break;
                    }
                }
            }
        }
    }

    if (final_state) *final_state = state;
    if (steps_out) *steps_out = (uint32_t)steps;
    if (tries_out) *tries_out = (uint32_t)tries;
    if (drift_out) *drift_out = max_drift;
    if (ncross_out) *ncross_out = ncross;
    if (rgba) {
        rgba[0] = (float)color[0];
        rgba[1] = (float)color[1];
        rgba[2] = (float)color[2];
        rgba[3] = 1.0f;
    }
    return term;
}

void orc_render_frame(const orc_camera *cam, const orc_frame_params *fp, const float *lut_in,
                      uint32_t stride_x, uint32_t stride_y, float *rgba, orc_state *final_states,
                      uint32_t *steps, uint8_t *term, double *drift, orc_frame_stats *stats,
                      int nthreads) {
    if (stride_x == 0) stride_x = 1;
    if (stride_y == 0) stride_y = 1;
    if (nthreads < 1) nthreads = 1;
    uint32_t nx = (fp->width + stride_x - 1) / stride_x;
    uint32_t ny = (fp->height + stride_y - 1) / stride_y;

    float *lut_own = NULL;
    const float *lut = lut_in;
    if (fp->shading && !lut) {
        lut_own = (float *)malloc((size_t)fp->lut_width * fp->lut_height * 4 * sizeof(float));
        orc_generate_blackbody_lut(fp->lut_width, fp->lut_height, fp->lut_max_temp, lut_own);
        lut = lut_own;
    }

    float disk_lut_own[512];
    orc_frame_params fp_local = *fp;
    if (fp->shading && fp->disk_profile == 1 && !fp->disk_lut) {
        orc_metric mk = orc_metric_make(ORC_KERR_BL, fp->mass, fp->spin);
        orc_generate_temperature_lut(mk.mass, mk.spin, 512, disk_lut_own);
        fp_local.disk_lut = disk_lut_own;
    }
    fp = &fp_local;

    uint64_t acc_steps = 0, acc_tries = 0, acc_cross = 0;
    uint64_t tc0 = 0, tc1 = 0, tc2 = 0, tc3 = 0, tc4 = 0;
    double maxd = 0.0;
    long long total = (long long)nx * ny;

#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads) if (nthreads > 1) \
    reduction(+ : acc_steps, acc_tries, acc_cross, tc0, tc1, tc2, tc3, tc4) reduction(max : maxd)
    for (long long k = 0; k < total; k++) {
        uint32_t jj = (uint32_t)(k / nx), ii = (uint32_t)(k % nx);
        orc_state fs;
        uint32_t st = 0, tr = 0, nc = 0;
        double dr = 0.0;
        float px[4];
        int t = orc_trace_pixel(cam, fp, lut, ii * stride_x, jj * stride_y, &fs, &st, &tr, &dr,
                                &nc, NULL, px);
        if (rgba) memcpy(&rgba[(size_t)k * 4], px, sizeof(px));
        if (final_states) final_states[k] = fs;
        if (steps) steps[k] = st;
        if (term) term[k] = (uint8_t)t;
        if (drift) drift[k] = dr;
        acc_steps += st;
        acc_tries += tr;
        acc_cross += nc;
        switch (t) {
        case 0: tc0++; break;
        case 1: tc1++; break;
        case 2: tc2++; break;
        case 3: tc3++; break;
        default: tc4++; break;
        }
        if (dr > maxd) maxd = dr;
    }
    if (stats) {
        stats->rays = (uint64_t)total;
        stats->accepted_steps = acc_steps;
        stats->rkf_tries = acc_tries;
        stats->term_count[0] = tc0;
        stats->term_count[1] = tc1;
        stats->term_count[2] = tc2;
        stats->term_count[3] = tc3;
        stats->term_count[4] = tc4;
        stats->crossings = acc_cross;
        stats->max_drift = maxd;
    }
    free(lut_own);
}
