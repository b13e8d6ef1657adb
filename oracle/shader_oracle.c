/*
 * shader_oracle.c -- f32 restatement of the reference's GPU march loops.
 * TEST INFRASTRUCTURE ONLY; see shader_oracle.h.
 *
 * Restated as written: every formula of the cited shader lines, in f32, in the
 * shader's evaluation order.  Deliberately fixed (the reference is not
 * reproducible there):
 *   - the shaders' sin / cos / pow / exp / log / acos / atan2 (precision left to the GPU by the
 *     GLSL ES and WGSL specifications): the f32 forms of ref_libm.c (the written-out f64 routine,
 *     rounded once), which the engine's shader-order kernels use too -- so those kernels return
 *     this file's bits, and FAST kernels are compared statistically;
 *   - WGSL star hash (compute.wgsl.ts:201-204): fract(sin(.) * 43758.5453) amplifies the last
 *     ulp of sin by 4e4: a different sin moves stars; `stars = 0` skips it for the FAST kernels;
 *   - GLSL noise textures (webgl-utils.ts:259-305 fills them with Math.random()): the caller
 *     supplies the two 256x256 R channels (orc_seeded_noise_rgba8 makes seeded ones);
 *     hash() = texture(u_noiseTex, (uv+0.5)/256).r with LINEAR/REPEAT is evaluated with f32
 *     bilinear weights (GPU samplers use fixed-point weights: not specified, not reproducible);
 *   - `turbulence >= 0` replaces the two noise() fetches of disk.ts:55 by that value
 *     (0.75 = a uniform texture of 1.0); ORC_GLSL_DITHER off = bNoise 0.
 * GLSL built-ins: normalize(v) = v / sqrt(dot(v,v)); smoothstep, clamp, mix, sign as in
 * the GLSL ES 3.00 specification.
 */
#include "shader_oracle.h"
#include "ref_libm.h"

#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------ helpers */
typedef struct { float x, y, z; } v3;

static float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 cross3(v3 a, v3 b) {
    v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    return r;
}
static v3 scale3(v3 a, float s) { v3 r = {a.x * s, a.y * s, a.z * s}; return r; }
static v3 add3(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static float length3(v3 a) { return sqrtf(dot3(a, a)); }
static v3 normalize3(v3 a) {
    float l = length3(a);
    v3 r = {a.x / l, a.y / l, a.z / l};
    return r;
}
static float smoothstepf(float e0, float e1, float x) {
    float t = clampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
static float signf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

/* ========================================================================== */
/* WGSL compute kernel                                                         */
/* ========================================================================== */
typedef struct { float x[4]; float p[4]; } ray32;

/* compute.wgsl.ts:28-32 */
static float wgsl_horizon(float M, float a) {
    float disc = M * M - a * a;
    if (disc < 0.0f) return M;
    return M + sqrtf(disc);
}
/* compute.wgsl.ts:34-40 */
static float wgsl_isco(float M, float a) {
    float rs = a / M;
    float absS = fabsf(clampf(rs, -0.999f, 0.999f));
    float z1 = 1.0f + orc_powf(1.0f - absS * absS, 1.0f / 3.0f) *
                          (orc_powf(1.0f + absS, 1.0f / 3.0f) + orc_powf(1.0f - absS, 1.0f / 3.0f));
    float z2 = sqrtf(3.0f * absS * absS + z1 * z1);
    return M * (3.0f + z2 - sqrtf((3.0f - z1) * (3.0f + z1 + 2.0f * z2)));
}

/* compute.wgsl.ts:42-120 */
static void wgsl_derivs(const ray32 *s, float M, float spin, float dx[4], float dp[4]) {
    float a = spin * M;
    float r = s->x[1], theta = s->x[2];
    float r2 = r * r, a2 = a * a;
    float sint = orc_sinf(theta), cost = orc_cosf(theta);
    float sin2 = fmaxf(sint * sint, 1e-12f);
    float cos2 = 1.0f - sin2;
    float sigma = r2 + a2 * cos2;
    float sigma2 = sigma * sigma;
    float delta = r2 - 2.0f * M * r + a2;

    float g_tt = -(1.0f + 2.0f * M * r / sigma);
    float g_tr = 2.0f * M * r / sigma;
    float g_rr = delta / sigma;
    float g_thth = 1.0f / sigma;
    float g_phph = 1.0f / (sigma * sin2);
    float g_rph = a / sigma;
    const float *p = s->p;

    dx[0] = g_tt * p[0] + g_tr * p[1];
    dx[1] = g_tr * p[0] + g_rr * p[1] + g_rph * p[3];
    dx[2] = g_thth * p[2];
    dx[3] = g_rph * p[1] + g_phph * p[3];

    float dsigma_dr = 2.0f * r;
    float dsigma_dth = -2.0f * a2 * sint * cost;
    float ddelta_dr = 2.0f * r - 2.0f * M;

    float dg_tt_dr = -(2.0f * M * (sigma - r * dsigma_dr)) / sigma2;
    float dg_tt_dth = (2.0f * M * r * dsigma_dth) / sigma2;
    float dg_tr_dr = -dg_tt_dr;
    float dg_tr_dth = -dg_tt_dth;
    float dg_rr_dr = (ddelta_dr * sigma - delta * dsigma_dr) / sigma2;
    float dg_rr_dth = -(delta * dsigma_dth) / sigma2;
    float dg_thth_dr = -dsigma_dr / sigma2;
    float dg_thth_dth = -dsigma_dth / sigma2;
    float dg_phph_dr = -dsigma_dr / (sigma2 * sin2);
    float dg_phph_dth = -(dsigma_dth * sin2 + sigma * orc_sinf(2.0f * theta)) / (sigma2 * sin2 * sin2);
    float dg_rph_dr = -(a * dsigma_dr) / sigma2;
    float dg_rph_dth = -(a * dsigma_dth) / sigma2;

    float dh_dr = 0.5f * (dg_tt_dr * p[0] * p[0] + dg_rr_dr * p[1] * p[1] + dg_thth_dr * p[2] * p[2] +
                          dg_phph_dr * p[3] * p[3] + 2.0f * dg_tr_dr * p[0] * p[1] +
                          2.0f * dg_rph_dr * p[1] * p[3]);
    float dh_dth = 0.5f * (dg_tt_dth * p[0] * p[0] + dg_rr_dth * p[1] * p[1] +
                           dg_thth_dth * p[2] * p[2] + dg_phph_dth * p[3] * p[3] +
                           2.0f * dg_tr_dth * p[0] * p[1] + 2.0f * dg_rph_dth * p[1] * p[3]);
    dp[0] = 0.0f;
    dp[1] = -dh_dr;
    dp[2] = -dh_dth;
    dp[3] = 0.0f;
}

/* compute.wgsl.ts:122-133 */
static ray32 wgsl_symplectic(const ray32 *s, float h, float M, float spin) {
    ray32 mid = *s;
    float dx[4], dp[4];
    for (int it = 0; it < 2; it++) {
        wgsl_derivs(&mid, M, spin, dx, dp);
        for (int k = 0; k < 4; k++) {
            float nx = s->x[k] + dx[k] * h, np = s->p[k] + dp[k] * h;
            mid.x[k] = (s->x[k] + nx) * 0.5f;
            mid.p[k] = (s->p[k] + np) * 0.5f;
        }
    }
    wgsl_derivs(&mid, M, spin, dx, dp);
    ray32 out;
    for (int k = 0; k < 4; k++) {
        out.x[k] = s->x[k] + dx[k] * h;
        out.p[k] = s->p[k] + dp[k] * h;
    }
    return out;
}

static void m4v4(const float *m, const float v[4], float out[4]) {
    for (int r = 0; r < 4; r++)
        out[r] = m[0 + r] * v[0] + m[4 + r] * v[1] + m[8 + r] * v[2] + m[12 + r] * v[3];
}

/* compute.wgsl.ts:147-258 */
uint32_t orc_wgsl_pixel(const orc_wgsl_params *P, uint32_t ix, uint32_t iy, float rgba[4]) {
    const float PI = 3.14159265f;
    float fw = (float)P->width, fh = (float)P->height;
    float jx = P->jitter[0] / fw, jy = P->jitter[1] / fh;
    float uvx = (float)ix / fw, uvy = (float)iy / fh;
    float ndcx = (uvx + jx) * 2.0f - 1.0f;
    float ndcy = (uvy + jy) * 2.0f - 1.0f;

    float clip[4] = {ndcx, -ndcy, 1.0f, 1.0f}, vt[4];
    m4v4(P->inv_proj, clip, vt);
    v3 vd = {vt[0] / vt[3], vt[1] / vt[3], vt[2] / vt[3]};
    vd = normalize3(vd);
    float vd4[4] = {vd.x, vd.y, vd.z, 0.0f}, wd4[4];
    m4v4(P->inv_view, vd4, wd4);
    v3 wd = {wd4[0], wd4[1], wd4[2]};
    wd = normalize3(wd);

    v3 cam = {P->position[0], P->position[1], P->position[2]};
    float r0 = length3(cam);
    float theta0 = orc_acosf(clampf(cam.y / r0, -1.0f, 1.0f));
    float phi0 = orc_atan2f(cam.z, cam.x);
    float st = orc_sinf(theta0), ct = orc_cosf(theta0), sp = orc_sinf(phi0), cp = orc_cosf(phi0);

    v3 e_r = {st * cp, ct, st * sp}, e_th = {ct * cp, -st, ct * sp}, e_ph = {-sp, 0.0f, cp};
    float pr_far = dot3(wd, e_r);
    float pth_far = dot3(wd, e_th) / r0;
    float safe_st = fmaxf(st, 1e-4f);
    float pph_far = dot3(wd, e_ph) / (r0 * safe_st);

    ray32 s;
    s.x[0] = 0.0f; s.x[1] = r0; s.x[2] = theta0; s.x[3] = phi0;
    s.p[0] = -1.0f; s.p[1] = pr_far; s.p[2] = pth_far * r0 * r0; s.p[3] = pph_far * r0 * r0 * st * st;

    float M = P->mass;
    float a = P->spin * M;
    float rh = wgsl_horizon(M, a);
    float isco = wgsl_isco(M, a);

    float color[3] = {0.0f, 0.0f, 0.0f};
    float alpha = 0.0f;
    uint32_t steps = 0;

    for (int i = 0; i < P->max_steps; i++) {
        float r = s.x[1];
        if (r < rh * 1.001f) break;
        if (r > 100.0f) { /* compute.wgsl.ts:199-206 */
            if (P->stars) {
                v3 vdir = {s.p[1], s.p[2] / r, s.p[3] / (r * safe_st)};
                vdir = normalize3(vdir);
                float sn = orc_sinf(vdir.x * 12.9898f + vdir.y * 78.233f + vdir.z * 45.164f) * 43758.5453f;
                float star = sn - floorf(sn); /* fract */
                if (star > 0.999f)
                    for (int c = 0; c < 3; c++) color[c] += 1.0f * (1.0f - alpha);
            }
            break;
        }
        float prev_theta = s.x[2];
        float h = clampf((r - rh) * 0.15f, 0.05f, 1.0f);
        s = wgsl_symplectic(&s, h, M, P->spin);
        steps++;
        float curr_theta = s.x[2];
        if ((prev_theta - PI * 0.5f) * (curr_theta - PI * 0.5f) <= 0.0f && r > isco && r < 30.0f) {
            float Omega = 1.0f / (orc_powf(r, 1.5f) + a);
            float u_t = 1.0f / sqrtf(fmaxf(1.0f - 2.0f * M / r - Omega * Omega * (r * r + a * a), 1e-4f));
            float u_phi = Omega * u_t;
            float g_factor = -s.p[0] / fmaxf(-(u_t * s.p[0] + u_phi * s.p[3]), 1e-4f);
            float artistic_T = (1.0f / orc_powf(fmaxf(r / isco, 1.0f), 0.75f)) * g_factor;
            float base[3] = {1.0f, 0.5f, 0.1f}, blue[3] = {0.5f, 0.7f, 1.0f}, red[3] = {1.0f, 0.2f, 0.0f};
            float bs = fmaxf(g_factor - 1.0f, 0.0f), rs = fmaxf(1.0f - g_factor, 0.0f) * 0.5f;
            float target_opacity = 0.6f * artistic_T;
            float g4 = orc_powf(g_factor, 4.0f);
            float mri_shear = orc_powf(r, -1.5f);
            float mri_sat = 1.0f + 0.0001f * orc_sinf(r * 100.0f * mri_shear);
            for (int c = 0; c < 3; c++) {
                float target = (base[c] + blue[c] * bs - red[c] * rs) * artistic_T * 4.0f;
                float I_em = target * target_opacity / fmaxf(g4, 1e-5f);
                float j_nu = I_em * mri_sat;
                float I_obs = g4 * j_nu;
                color[c] += I_obs * (1.0f - alpha);
            }
            alpha += target_opacity * mri_sat;
        }
        if (alpha > 0.99f) break;
    }
    rgba[0] = color[0]; rgba[1] = color[1]; rgba[2] = color[2]; rgba[3] = 1.0f;
    return steps;
}

void orc_wgsl_frame(const orc_wgsl_params *p, uint32_t sx, uint32_t sy, float *rgba,
                    uint32_t *steps, int nthreads) {
    if (sx == 0) sx = 1;
    if (sy == 0) sy = 1;
    if (nthreads < 1) nthreads = 1;
    uint32_t nx = (p->width + sx - 1) / sx, ny = (p->height + sy - 1) / sy;
    long long total = (long long)nx * ny;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads) if (nthreads > 1)
    for (long long k = 0; k < total; k++) {
        float px[4];
        uint32_t st = orc_wgsl_pixel(p, (uint32_t)(k % nx) * sx, (uint32_t)(k / nx) * sy, px);
        if (rgba) memcpy(&rgba[k * 4], px, sizeof px);
        if (steps) steps[k] = st;
    }
}

/* ========================================================================== */
/* GLSL fragment march                                                         */
/* ========================================================================== */
#define GL_PI 3.14159265359f
#define GL_MAX_DIST 10000.0f /* physics.config.ts:60 */
#define GL_MIN_STEP 0.01f    /* :61 */
#define GL_MAX_STEP 1.2f     /* :62 */
#define GL_HORIZON_THRESHOLD 1.15f /* :63 */

/* chunks/metric.ts:13-37 */
static float gl_horizon(float M, float a) { return M + sqrtf(fmaxf(0.0f, M * M - a * a)); }
static float gl_isco(float M, float a) {
    float rs = a / M;
    float absS = fabsf(clampf(rs, -0.9999f, 0.9999f));
    float z1 = 1.0f + orc_powf(1.0f - absS * absS, 1.0f / 3.0f) *
                          (orc_powf(1.0f + absS, 1.0f / 3.0f) + orc_powf(1.0f - absS, 1.0f / 3.0f));
    float z2 = sqrtf(3.0f * absS * absS + z1 * z1);
    float signOfA = signf(a);
    if (signOfA == 0.0f) signOfA = 1.0f;
    return M * (3.0f + z2 - signOfA * sqrtf((3.0f - z1) * (3.0f + z1 + 2.0f * z2)));
}
static float gl_photon_sphere(float M, float a) {
    float a_star = clampf(a / M, -0.9999f, 0.9999f);
    float arg = clampf(-a_star, -1.0f, 1.0f);
    float theta = (2.0f / 3.0f) * orc_acosf(arg);
    return 2.0f * M * (1.0f + orc_cosf(theta));
}

/* chunks/metric.ts:96-149 */
static v3 gl_kerr_accel(v3 p, v3 v, float M, float a, float *omega) {
    float a2 = a * a;
    float rho2 = dot3(p, p);
    float diff = rho2 - a2;
    float disc = diff * diff + 4.0f * a2 * p.y * p.y;
    float r2 = 0.5f * (diff + sqrtf(fmaxf(0.0f, disc)));
    float r_k = sqrtf(fmaxf(1e-8f, r2));

    float sigma = r2 + a2 * (p.y * p.y / fmaxf(1e-8f, r2));
    v3 L = cross3(p, v);
    float Ly = L.y;
    float Ly_eff = Ly - a;
    float L2_eff = Ly_eff * Ly_eff + (dot3(L, L) - Ly * Ly);

    float r_inv = 1.0f / r_k;
    float r2_inv = r_inv * r_inv;
    float r4_inv = r2_inv * r2_inv;
    float sigma_ratio = r2 / fmaxf(1e-8f, sigma);
    v3 n = normalize3(p);
    v3 r_hat = {-n.x, -n.y, -n.z};
    v3 acc = scale3(r_hat, M * r2_inv * sigma_ratio + 3.0f * M * fmaxf(0.0f, L2_eff) * r4_inv * sigma_ratio);

    float r3_p_a2r = r_k * r2 + a2 * r_k;
    float drag = 2.0f * M * a / fmaxf(1e-8f, r3_p_a2r);
    v3 yhat = {0.0f, 1.0f, 0.0f};
    acc = add3(acc, scale3(cross3(yhat, v), drag));
    *omega = 2.0f * M * a / fmaxf(1e-8f, r3_p_a2r);
    return acc;
}

/* fragment.glsl.ts:204 -- the expression advanced-physics.test.ts:12-14 mirrors */
static float gl_redshift_potential(float rs, float r) { return sqrtf(fmaxf(0.0f, 1.0f - rs / r)); }
/* fragment.glsl.ts:265 -- advanced-physics.test.ts:17-20 */
static float gl_ergosphere_radius(float M, float a, float cosTheta) {
    return M + sqrtf(fmaxf(0.0f, M * M - a * a * cosTheta * cosTheta));
}
/* chunks/disk.ts:95 (Doppler beaming of the disk emission) */
static float gl_beaming(float delta) { return fmaxf(0.01f, orc_powf(delta, 3.5f)); }
/* chunks/disk.ts:78-93: Keplerian emitter, equatorial metric, u^t, photon L, delta */
static float gl_disk_delta(float M, float a, float spin, float sampleR, float L_photon) {
    float r2 = sampleR * sampleR;
    float sqrt_M = sqrtf(M);
    float signSpin = signf(spin + 1e-8f);
    float Omega = (signSpin * sqrt_M) / (sampleR * sqrtf(sampleR) + a * sqrt_M);
    float g_tt = -(1.0f - 2.0f * M / sampleR);
    float g_tphi = -2.0f * M * a / sampleR;
    float g_phiphi = r2 + a * a + 2.0f * M * a * a / sampleR;
    float u_t_sq = -(g_tt + 2.0f * Omega * g_tphi + Omega * Omega * g_phiphi);
    float u_t = 1.0f / sqrtf(fmaxf(1e-6f, u_t_sq));
    return 1.0f / fmaxf(0.01f, u_t * (1.0f - Omega * L_photon));
}
/* fragment.glsl.ts:176-191: the Velocity-Verlet update (advanced-physics.test.ts:196-218 runs the
 * same two lines on a harmonic oscillator) */
static v3 gl_verlet_position(v3 p, v3 v, v3 accel, float dt) {
    return add3(p, add3(scale3(v, dt), scale3(scale3(scale3(accel, 0.5f), dt), dt)));
}
static v3 gl_verlet_velocity(v3 v, v3 accel, v3 accel_new, float dt) {
    return add3(v, scale3(scale3(add3(accel, accel_new), 0.5f), dt));
}

/* chunks/blackbody.ts:9-34 */
static void gl_blackbody(float temp, float rgb[3]) {
    float t = fmaxf(temp, 1.0f) / 100.0f;
    float r, g, b;
    if (t <= 66.0f) {
        r = 255.0f;
        g = 99.4708025861f * orc_logf(t) - 161.1195681661f;
        if (t <= 19.0f) b = 0.0f;
        else b = 138.5177312231f * orc_logf(t - 10.0f) - 305.0447927307f;
    } else {
        r = 329.698727446f * orc_powf(t - 60.0f, -0.1332047592f);
        g = 288.1221695283f * orc_powf(t - 60.0f, -0.0755148492f);
        b = 255.0f;
    }
    rgb[0] = orc_powf(fmaxf(r / 255.0f, 0.0f), 2.2f);
    rgb[1] = orc_powf(fmaxf(g / 255.0f, 0.0f), 2.2f);
    rgb[2] = orc_powf(fmaxf(b / 255.0f, 0.0f), 2.2f);
}

/* chunks/common.ts:44-47 : mat2(c,-s,s,c), and `v.xy *= m` is row-vector * matrix */
static void gl_rot_apply(float ang, float *x, float *y) {
    float s = orc_sinf(ang), c = orc_cosf(ang);
    float nx = *x * c + *y * (-s);
    float ny = *x * s + *y * c;
    *x = nx;
    *y = ny;
}

/* ---- chunks/noise.ts ---- */
static float gl_fract(float x) { return x - floorf(x); }
static float gl_mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }

static float tex_r(const uint8_t *t, int x, int y) { /* REPEAT wrap, UNORM8 */
    return (float)t[(size_t)(y & 255) * 256u + (size_t)(x & 255)] / 255.0f;
}
/* texture(u_noiseTex, (uv + 0.5) / 256.0).r, LINEAR + REPEAT (GLSL ES 3.00 / GL ES 3.0 8.14) */
static float gl_hash_uv(const orc_glsl_params *U, float uvx, float uvy) {
    float s = (uvx + 0.5f) / 256.0f, t = (uvy + 0.5f) / 256.0f;
    float u = s * 256.0f - 0.5f, v = t * 256.0f - 0.5f;
    float fu = floorf(u), fv = floorf(v);
    float a = u - fu, b = v - fv;
    int i0 = (int)fmodf(fu, 256.0f), j0 = (int)fmodf(fv, 256.0f); /* & 255 wraps negatives */
    const uint8_t *T = U->noise_r;
    float t00 = tex_r(T, i0, j0), t10 = tex_r(T, i0 + 1, j0);
    float t01 = tex_r(T, i0, j0 + 1), t11 = tex_r(T, i0 + 1, j0 + 1);
    return (1.0f - a) * (1.0f - b) * t00 + a * (1.0f - b) * t10 + (1.0f - a) * b * t01 + a * b * t11;
}
static float gl_hash(const orc_glsl_params *U, v3 p) { /* noise.ts:3-9 */
    return gl_hash_uv(U, p.x + p.z * 37.0f, p.y + p.z * 37.0f);
}
static float gl_noise(const orc_glsl_params *U, v3 p) { /* noise.ts:11-21 */
    v3 i = {floorf(p.x), floorf(p.y), floorf(p.z)};
    v3 f = {gl_fract(p.x), gl_fract(p.y), gl_fract(p.z)};
    f.x = f.x * f.x * (3.0f - 2.0f * f.x);
    f.y = f.y * f.y * (3.0f - 2.0f * f.y);
    f.z = f.z * f.z * (3.0f - 2.0f * f.z);
#define H(dx, dy, dz) gl_hash(U, (v3){i.x + dx, i.y + dy, i.z + dz})
    float r = gl_mix(gl_mix(gl_mix(H(0, 0, 0), H(1, 0, 0), f.x), gl_mix(H(0, 1, 0), H(1, 1, 0), f.x), f.y),
                     gl_mix(gl_mix(H(0, 0, 1), H(1, 0, 1), f.x), gl_mix(H(0, 1, 1), H(1, 1, 1), f.x), f.y),
                     f.z);
#undef H
    return r;
}
static float gl_fbm(const orc_glsl_params *U, v3 p) { /* noise.ts:23-33 */
    float f = 0.0f, amp = 0.5f;
    for (int i = 0; i < 4; i++) {
        f += amp * gl_noise(U, p);
        p = scale3(p, 2.0f);
        amp *= 0.5f;
    }
    return f;
}

/* chunks/blackbody.ts:36-46 */
static void gl_star_color(float bv, float c[3]) {
    float t = clampf(bv, -0.4f, 2.0f);
    if (t < 0.0f) { c[0] = 0.6f; c[1] = 0.7f; c[2] = 1.0f; }
    else if (t < 0.3f) { c[0] = 0.85f; c[1] = 0.88f; c[2] = 1.0f; }
    else if (t < 0.6f) { c[0] = 1.0f; c[1] = 0.96f; c[2] = 0.9f; }
    else if (t < 1.0f) { c[0] = 1.0f; c[1] = 0.85f; c[2] = 0.6f; }
    else { c[0] = 1.0f; c[1] = 0.6f; c[2] = 0.4f; }
}

/* chunks/background.ts:3-30 */
static void gl_starfield(const orc_glsl_params *U, v3 dir, float stars[3]) {
    stars[0] = stars[1] = stars[2] = 0.0f;
    v3 cell = {floorf(dir.x * 200.0f), floorf(dir.y * 200.0f), floorf(dir.z * 200.0f)};
    float starNoise = gl_hash(U, cell);
    if (starNoise > 0.998f) {
        float brightness = orc_powf(starNoise, 10.0f) * 2.0f;
        float bv = gl_hash(U, (v3){cell.x + 127.1f, cell.y + 127.1f, cell.z + 127.1f}) * 2.4f - 0.4f;
        float twinkle = 0.85f + 0.15f * orc_sinf(U->time * (3.0f + gl_hash(U, (v3){cell.x + 73.7f, cell.y + 73.7f, cell.z + 73.7f}) * 2.0f));
        float sc[3];
        gl_star_color(bv, sc);
        for (int c = 0; c < 3; c++) stars[c] = sc[c] * brightness * twinkle;
    }
    cell = (v3){floorf(dir.x * 500.0f), floorf(dir.y * 500.0f), floorf(dir.z * 500.0f)};
    starNoise = gl_hash(U, cell);
    if (starNoise > 0.996f) {
        float brightness = orc_powf(starNoise, 20.0f) * 1.5f;
        float bv = gl_hash(U, (v3){cell.x + 217.3f, cell.y + 217.3f, cell.z + 217.3f}) * 2.4f - 0.4f;
        float sc[3];
        gl_star_color(bv, sc);
        for (int c = 0; c < 3; c++) stars[c] += sc[c] * brightness;
    }
    float tt = U->time * 0.01f;
    float nebula = gl_fbm(U, (v3){dir.x * 2.0f + tt, dir.y * 2.0f + tt, dir.z * 2.0f + tt}) * 0.03f;
    float ln = fabsf(nebula); /* length(float) */
    stars[0] += nebula * 0.2f + 0.05f * ln;
    stars[1] += nebula * 0.3f + 0.02f * ln;
    stars[2] += nebula * 0.5f + 0.05f * ln;
}

/* chunks/disk.ts:16-115 */
static void gl_sample_disk(const orc_glsl_params *U, v3 p, v3 p_prev, v3 v, float isco, float M,
                           float a, float dt, float color[3], float *alpha) {
    if (!(U->show_redshift < 0.5f)) return;
    int crossed = (p_prev.y * p.y < 0.0f);
    v3 sp = p;
    if (crossed) {
        float t = fabsf(p_prev.y) / fmaxf(0.0001f, fabsf(p_prev.y) + fabsf(p.y));
        sp.x = p_prev.x * (1.0f - t) + p.x * t; /* mix */
        sp.y = p_prev.y * (1.0f - t) + p.y * t;
        sp.z = p_prev.z * (1.0f - t) + p.z * t;
    }
    float sampleR = length3(sp);
    float effH = fminf(U->disk_scale_height, 0.45f); /* physics.config.ts:49 */
    float diskHeight = sampleR * effH;
    float diskInner = isco;
    float diskOuter = fmaxf(M * U->disk_size, diskInner * 1.1f);
    if (!((fabsf(sp.y) < diskHeight || crossed) && sampleR > diskInner && sampleR < diskOuter)) return;

    float turbulence = U->turbulence;
    if (turbulence < 0.0f) { /* disk.ts:43-55 */
        float sqrt_M_phase = sqrtf(M);
        float signSpinPhase = signf(U->spin + 1e-8f);
        float OmegaPhase = (signSpinPhase * sqrt_M_phase) / (sampleR * sqrtf(sampleR) + a * sqrt_M_phase);
        float rotAngle = OmegaPhase * U->time * 0.12f * 10.0f;
        v3 np = sp;
        /* mat2(cos, -sin, sin, cos); noiseP.xz *= rotPhase (row vector * matrix) */
        float cs = orc_cosf(rotAngle), sn = orc_sinf(rotAngle);
        float nx = np.x * cs + np.z * (-sn), nz = np.x * sn + np.z * cs;
        np.x = nx;
        np.z = nz;
        np = scale3(np, 0.75f);
        turbulence = gl_noise(U, np) * 0.5f + gl_noise(U, scale3(np, 2.5f)) * 0.25f;
    }
    float samplesDiskHeight = sampleR * effH;
    float heightFalloff = orc_expf(-fabsf(sp.y) / fmaxf(0.001f, samplesDiskHeight * 0.25f));
    float radialFalloff = smoothstepf(diskOuter, diskInner, sampleR);
    float baseDensity = turbulence * heightFalloff * radialFalloff;
    if (!(baseDensity > 0.001f)) return;

    float L_photon = p.z * v.x - p.x * v.z;
    float delta = gl_disk_delta(M, a, U->spin, sampleR, L_photon);
    float beaming = (U->features & ORC_GLSL_DOPPLER) ? gl_beaming(delta) : 1.0f;
    float isco_r = clampf(isco / sampleR, 0.0f, 1.0f);
    float nt_factor = fmaxf(0.0f, 1.0f - sqrtf(isco_r));
    float radialTempGradient = orc_powf(isco_r, 0.75f) * orc_powf(nt_factor, 0.25f);
    float temperature = U->disk_temp * radialTempGradient * delta;
    float bb[3];
    gl_blackbody(temperature, bb);
    float density = baseDensity * U->disk_density * 0.12f * dt;
    for (int c = 0; c < 3; c++) color[c] += bb[c] * beaming * density * (1.0f - *alpha);
    *alpha += density;
}

/* chunks/disk.ts:117-155 */
static void gl_sample_jets(const orc_glsl_params *U, v3 p, v3 v, float rh, float dt, float color[3],
                           float *alpha) {
    float jetVerticalPos = fabsf(p.y);
    if (!(jetVerticalPos > rh * 1.8f && jetVerticalPos < GL_MAX_DIST * 0.8f)) return;
    float jetRadialDist = sqrtf(p.x * p.x + p.z * p.z);
    float jetWidth = 1.0f + jetVerticalPos * 0.15f;
    if (!(jetRadialDist < jetWidth * 2.0f)) return;
    float radialFalloff = orc_expf(-(jetRadialDist * jetRadialDist) / (jetWidth * 0.5f));
    float lengthFalloff = orc_expf(-jetVerticalPos * 0.05f);
    float flowCombined = p.y * 2.0f - U->time * 8.0f;
    v3 uvJet = {p.x, flowCombined, p.z};
    float noiseVal = gl_noise(U, scale3(uvJet, 0.5f)) * 0.6f + gl_noise(U, scale3(uvJet, 1.5f)) * 0.4f;
    float jetDensity = radialFalloff * lengthFalloff * fmaxf(0.0f, noiseVal - 0.2f);
    if (!(jetDensity > 0.001f)) return;
    float jetVel = 0.92f * signf(p.y);
    v3 jetVelVec = {0.0f, jetVel, 0.0f};
    v3 nv = normalize3(jetVelVec);
    float cosThetaJet = dot3(nv, (v3){-v.x, -v.y, -v.z});
    float betaJet = fabsf(jetVel);
    float gammaJet = 1.0f / sqrtf(1.0f - betaJet * betaJet);
    float deltaJet = 1.0f / (gammaJet * (1.0f - betaJet * cosThetaJet));
    float beamingJet = orc_powf(deltaJet, 3.5f);
    const float base[3] = {0.4f, 0.7f, 1.0f};
    for (int c = 0; c < 3; c++) color[c] += base[c] * jetDensity * 0.05f * beamingJet * dt * (1.0f - *alpha);
    *alpha += jetDensity * 0.05f * dt;
}

static float aces(float c) { /* chunks/common.ts:50-57 */
    const float A = 2.51f, B = 0.03f, C = 2.43f, D = 0.59f, E = 0.14f;
    return clampf((c * (A * c + B)) / (c * (C * c + D) + E), 0.0f, 1.0f);
}

static v3 gl_qrot(const float q[4], v3 v) { /* common.ts:74-76 */
    v3 qv = {q[0], q[1], q[2]};
    v3 t = add3(cross3(qv, v), scale3(v, q[3]));
    return add3(v, scale3(cross3(qv, t), 2.0f));
}

/* fragment.glsl.ts:40-334 (the whole main()) */
uint32_t orc_glsl_pixel(const orc_glsl_params *U, uint32_t ix, uint32_t iy, float rgba[4]) {
    const uint32_t F = U->features;
    float resx = (float)U->width, resy = (float)U->height;
    float minRes = fminf(resx, resy);
    /* gl_FragCoord = pixel centre, origin bottom-left; iy counts image rows from the top */
    float fcx = (float)ix + 0.5f, fcy = (float)(U->height - 1u - iy) + 0.5f;
    float uvx = (fcx - 0.5f * resx) / minRes, uvy = (fcy - 0.5f * resy) / minRes;
    rgba[3] = 1.0f;
    if (U->debug > 0.5f) {
        rgba[0] = uvx + 0.5f;
        rgba[1] = uvy + 0.5f;
        rgba[2] = 0.0f;
        return 0;
    }

    v3 ro, rd;
    v3 cp = {U->cam_pos[0], U->cam_pos[1], U->cam_pos[2]};
    if (length3(cp) > 0.001f) {
        ro = cp;
        rd = gl_qrot(U->cam_quat, normalize3((v3){uvx, uvy, 1.2f}));
    } else {
        ro = (v3){0.0f, 0.0f, -U->zoom};
        rd = normalize3((v3){uvx, uvy, 1.5f});
        float ax = (U->mouse[1] - 0.5f) * GL_PI, ay = (U->mouse[0] - 0.5f) * GL_PI * 2.0f;
        gl_rot_apply(ax, &ro.y, &ro.z);
        gl_rot_apply(ax, &rd.y, &rd.z);
        gl_rot_apply(ay, &ro.x, &ro.z);
        gl_rot_apply(ay, &rd.x, &rd.z);
    }

    float M = U->mass;
    float rs = M * 2.0f;
    float a = U->spin * M;
    float rh = gl_horizon(M, a);
    float rph = gl_photon_sphere(M, a);
    float isco = gl_isco(M, a);
    float absA = fabsf(U->spin);

    if (U->quality == 0) { /* fragment.glsl.ts:76-88 */
        float bg[3];
        gl_starfield(U, rd, bg);
        float d = length3(cross3(ro, rd));
        float shadow = smoothstepf(rh * 1.2f, rh * 0.9f, d);
        float glow = orc_expf(-fabsf(d - rph) * 12.0f) * 0.8f;
        const float glowCol[3] = {0.3f * glow, 0.6f * glow, 1.0f * glow};
        float diskMask = smoothstepf(isco * 2.0f, isco * 1.0f, d) * (1.0f - smoothstepf(isco * 1.0f, isco * 0.8f, d));
        const float diskCol[3] = {1.0f * diskMask * 0.6f, 0.7f * diskMask * 0.6f, 0.3f * diskMask * 0.6f};
        for (int c = 0; c < 3; c++)
            rgba[c] = orc_powf(bg[c] * (1.0f - shadow) + glowCol[c] + diskCol[c], 0.4545f);
        return 0;
    }

    v3 p = ro, v = rd;
    if (length3(ro) < rh * 1.5f) {
        ro = scale3(scale3(normalize3(ro), rh), 1.5f); /* normalize(ro) * rh * 1.5 */
        p = ro;
    }
    float color[3] = {0.0f, 0.0f, 0.0f};
    float alpha = 0.0f;
    int hitHorizon = 0;
    float maxRedshift = 0.0f;

    /* blue-noise dither: texture(u_blueNoiseTex, gl_FragCoord.xy / 256).r, NEAREST + REPEAT */
    float bNoise = 0.0f;
    if (F & ORC_GLSL_DITHER) bNoise = tex_r(U->blue_r, (int)floorf(fcx), (int)floorf(fcy));
    p = add3(p, scale3(scale3(v, bNoise), GL_MIN_STEP));

    int photonCrossings = 0;
    float prevY = p.y;
    float impactParam = length3(cross3(ro, rd));
    int redshiftInitialized = 0;
    int maxSteps = (int)fminf((float)U->max_ray_steps, 500.0f);
    v3 p_prev = p;
    if (impactParam < rh * 0.9f) hitHorizon = 1;
    uint32_t steps = 0;

    for (int i = 0; i < maxSteps; i++) {
        p_prev = p;
        float r = length3(p);
        if (r < rh * GL_HORIZON_THRESHOLD) {
            hitHorizon = 1;
            break;
        }
        if (r > GL_MAX_DIST) break;

        float distFactor = 1.0f + r * 0.05f;
        float dt = clampf((r - rh) * 0.1f * distFactor, GL_MIN_STEP, GL_MAX_STEP * distFactor);
        if (r > 30.0f) {
            float farBoost = (r - 30.0f) * 0.08f;
            dt = fmaxf(dt, GL_MIN_STEP + farBoost);
            dt = fminf(dt, GL_MAX_STEP * 2.5f);
        }
        float sphereProx = fabsf(r - rph);
        dt = fminf(dt, GL_MIN_STEP + sphereProx * 0.15f);
        float hRefinement = smoothstepf(0.2f, 0.0f, fabsf(p.y));
        float currentDt = dt * (1.0f - hRefinement * 0.7f);

        v3 accel = {0.0f, 0.0f, 0.0f};
        if (F & ORC_GLSL_LENSING) {
            float omega;
            accel = scale3(gl_kerr_accel(p, v, M, a, &omega), U->lensing_strength);
            gl_rot_apply(omega * currentDt, &v.x, &v.z); /* ZAMO twist of the velocity */
        }
        p = gl_verlet_position(p, v, accel, currentDt);
        float r_new = length3(p);
        if ((F & ORC_GLSL_LENSING) && alpha < 0.95f) {
            float om2;
            v3 accel_new = scale3(gl_kerr_accel(p, v, M, a, &om2), U->lensing_strength);
            v = gl_verlet_velocity(v, accel, accel_new, currentDt);
        }
        v = normalize3(v);
        steps++;

        if (prevY * p.y < 0.0f && r_new < rph * 2.0f && r_new > rh)
            photonCrossings = photonCrossings + 1 < 3 ? photonCrossings + 1 : 3;
        if (U->show_redshift > 0.5f) {
            float potential = gl_redshift_potential(rs, r_new);
            if (!redshiftInitialized) {
                maxRedshift = potential;
                redshiftInitialized = 1;
            } else
                maxRedshift = fminf(maxRedshift, potential);
        }
        prevY = p.y;

        if (F & ORC_GLSL_DISK) {
            gl_sample_disk(U, p, p_prev, v, isco, M, a, currentDt, color, &alpha);
            if (alpha > 0.99f) break;
        }
        /* jets take the un-refined dt and the pre-step r (fragment.glsl.ts:219) */
        if ((F & ORC_GLSL_JETS) && (F & ORC_GLSL_DISK)) gl_sample_jets(U, p, v, rh, dt, color, &alpha);
    }

    if ((F & ORC_GLSL_REDSHIFT) && U->show_redshift > 0.5f) { /* fragment.glsl.ts:224-237 */
        float val = hitHorizon ? 0.0f : maxRedshift;
        float h[3] = {0.0f, 0.0f, 0.0f};
        const float c1[3] = {1.0f, 0.0f, 0.0f}, c2[3] = {1.0f, 1.0f, 0.0f}, c3[3] = {0.0f, 0.0f, 1.0f};
        float t1 = smoothstepf(0.0f, 0.3f, val), t2 = smoothstepf(0.3f, 0.7f, val), t3 = smoothstepf(0.7f, 1.0f, val);
        for (int c = 0; c < 3; c++) {
            h[c] = gl_mix(0.0f, c1[c], t1);
            h[c] = gl_mix(h[c], c2[c], t2);
            h[c] = gl_mix(h[c], c3[c], t3);
            rgba[c] = h[c];
        }
        return steps;
    }

    float background[3] = {0.0f, 0.0f, 0.0f};
    if (F & ORC_GLSL_STARS) gl_starfield(U, v, background);

    float photonColor = 0.0f;
    if ((F & ORC_GLSL_PHOTON_GLOW) && !hitHorizon) { /* fragment.glsl.ts:246-258 */
        float distToPhotonRing = fabsf(length3(p) - rph);
        float directRing = orc_expf(-distToPhotonRing * 40.0f) * 1.8f * U->lensing_strength;
        float higherOrderRing = 0.0f;
        if (photonCrossings > 0) {
            float ringSharpness = 60.0f + (float)photonCrossings * 30.0f;
            float ringBrightness = orc_expf(-(float)photonCrossings * 1.0f) * 1.2f;
            higherOrderRing = orc_expf(-distToPhotonRing * ringSharpness) * ringBrightness * U->lensing_strength;
        }
        photonColor = 1.0f * (directRing + higherOrderRing);
    }

    float ergo[3] = {0.0f, 0.0f, 0.0f};
    if (absA > 0.1f && !hitHorizon) { /* fragment.glsl.ts:261-268 */
        float rFinal = length3(p);
        float cosTheta = p.y / fmaxf(rFinal, 0.001f);
        float r_ergo = gl_ergosphere_radius(M, a, cosTheta);
        float ergoGlow = orc_expf(-fabsf(rFinal - r_ergo) * 20.0f) * 0.35f * absA;
        ergo[0] = 0.3f * ergoGlow;
        ergo[1] = 0.35f * ergoGlow;
        ergo[2] = 0.9f * ergoGlow;
    }
    if (hitHorizon) background[0] = background[1] = background[2] = 0.0f;

    float fin[3];
    for (int c = 0; c < 3; c++)
        fin[c] = background[c] * (1.0f - alpha) + color[c] + photonColor * (1.0f - alpha) + ergo[c] * (1.0f - alpha);

    if (U->show_kerr_shadow > 0.5f) { /* fragment.glsl.ts:279-324 */
        v3 spin_axis = {0.0f, 1.0f, 0.0f};
        v3 cam_dir = normalize3(ro);
        v3 sky_right = normalize3(cross3(spin_axis, cam_dir));
        v3 sky_up = cross3(cam_dir, sky_right);
        v3 impact_vec = scale3(cross3(cam_dir, rd), length3(ro));
        float alpha_s = -dot3(impact_vec, sky_up);
        float beta_s = dot3(impact_vec, sky_right);
        float minDist = 1e10f;
        int count = (int)U->shadow_count;
        for (int j = 0; j < 63; j++) {
            if (j >= count - 1) break;
            float p1x = U->shadow_curve[j][0], p1y = U->shadow_curve[j][1];
            float p2x = U->shadow_curve[j + 1][0], p2y = U->shadow_curve[j + 1][1];
            float pax = alpha_s - p1x, pay = beta_s - p1y, bax = p2x - p1x, bay = p2y - p1y;
            float h = clampf((pax * bax + pay * bay) / (bax * bax + bay * bay), 0.0f, 1.0f);
            float dx = pax - bax * h, dy = pay - bay * h;
            minDist = fminf(minDist, sqrtf(dx * dx + dy * dy));
        }
        if (count > 2) {
            float pfx = U->shadow_curve[0][0], pfy = U->shadow_curve[0][1];
            float plx = U->shadow_curve[count - 1][0], ply = U->shadow_curve[count - 1][1];
            float pax = alpha_s - plx, pay = beta_s - ply, bax = pfx - plx, bay = pfy - ply;
            float h = clampf((pax * bax + pay * bay) / (bax * bax + bay * bay), 0.0f, 1.0f);
            float dx = pax - bax * h, dy = pay - bay * h;
            minDist = fminf(minDist, sqrtf(dx * dx + dy * dy));
        }
        float thickness = M * 0.045f;
        if (minDist < thickness) {
            float edge = smoothstepf(thickness, thickness * 0.5f, minDist);
            const float green[3] = {0.0f, 1.0f, 0.0f};
            for (int c = 0; c < 3; c++) fin[c] = gl_mix(fin[c], green[c], 1.0f * edge);
        }
    }

    for (int c = 0; c < 3; c++) {
        float f = fin[c];
        if (U->tone_map) f = orc_powf(fmaxf(aces(f), 0.0f), 0.4545f);
        rgba[c] = f;
    }
    return steps;
}

/* xorshift32 byte stream, value = floor(u * 255) with u in [0, 1) as createNoiseTexture does */
void orc_seeded_noise_rgba8(uint32_t seed, uint32_t size, uint8_t *rgba) {
    uint32_t x = seed ? seed : 0x9E3779B9u;
    const size_t n = (size_t)size * size * 4u;
    for (size_t i = 0; i < n; i++) {
        x ^= x << 13;
        x ^= x >> 17;
        x ^= x << 5;
        double u = (double)(x >> 8) / 16777216.0;
        rgba[i] = (uint8_t)floor(u * 255.0);
    }
}

void orc_glsl_frame(const orc_glsl_params *p, uint32_t sx, uint32_t sy, float *rgba,
                    uint32_t *steps, int nthreads) {
    if (sx == 0) sx = 1;
    if (sy == 0) sy = 1;
    if (nthreads < 1) nthreads = 1;
    uint32_t nx = (p->width + sx - 1) / sx, ny = (p->height + sy - 1) / sy;
    long long total = (long long)nx * ny;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads) if (nthreads > 1)
    for (long long k = 0; k < total; k++) {
        float px[4];
        uint32_t st = orc_glsl_pixel(p, (uint32_t)(k % nx) * sx, (uint32_t)(k / nx) * sy, px);
        if (rgba) memcpy(&rgba[k * 4], px, sizeof px);
        if (steps) steps[k] = st;
    }
}


/* ========================================================================== */
/* Test hooks: the building blocks above, callable one at a time, so that CPU  */
/* tests can hold them to known answers WITHOUT going through a march:         */
/*  - the reference's own tests of the shader expressions                      */
/*    (src/__tests__/physics/advanced-physics.test.ts),                        */
/*  - the pinned f64 oracle (gravitas_oracle.c), which the WGSL kernel's       */
/*    get_derivatives claims to equal formula for formula                      */
/*    (compute.wgsl.ts:42-120 vs kerr.rs:412-499),                             */
/*  - closed forms.                                                            */
/* ========================================================================== */
void orc_hook_wgsl_derivs(const float x[4], const float p[4], float M, float spin, float dx[4], float dp[4]) {
    ray32 s;
    memcpy(s.x, x, sizeof s.x);
    memcpy(s.p, p, sizeof s.p);
    wgsl_derivs(&s, M, spin, dx, dp);
}
void orc_hook_wgsl_step(const float x[4], const float p[4], float h, float M, float spin, float ox[4], float op[4]) {
    ray32 s;
    memcpy(s.x, x, sizeof s.x);
    memcpy(s.p, p, sizeof s.p);
    ray32 o = wgsl_symplectic(&s, h, M, spin);
    memcpy(ox, o.x, sizeof o.x);
    memcpy(op, o.p, sizeof o.p);
}
float orc_hook_wgsl_horizon(float M, float a) { return wgsl_horizon(M, a); }
float orc_hook_wgsl_isco(float M, float a) { return wgsl_isco(M, a); }
void orc_hook_glsl_accel(const float p[3], const float v[3], float M, float a, float acc[3], float *omega) {
    v3 r = gl_kerr_accel((v3){p[0], p[1], p[2]}, (v3){v[0], v[1], v[2]}, M, a, omega);
    acc[0] = r.x;
    acc[1] = r.y;
    acc[2] = r.z;
}
float orc_hook_glsl_horizon(float M, float a) { return gl_horizon(M, a); }
float orc_hook_glsl_isco(float M, float a) { return gl_isco(M, a); }
float orc_hook_glsl_photon_sphere(float M, float a) { return gl_photon_sphere(M, a); }
float orc_hook_glsl_redshift_potential(float rs, float r) { return gl_redshift_potential(rs, r); }
float orc_hook_glsl_ergosphere_radius(float M, float a, float c) { return gl_ergosphere_radius(M, a, c); }
float orc_hook_glsl_beaming(float delta) { return gl_beaming(delta); }
float orc_hook_glsl_disk_delta(float M, float a, float spin, float r, float L) { return gl_disk_delta(M, a, spin, r, L); }
void orc_hook_glsl_blackbody(float temp, float rgb[3]) { gl_blackbody(temp, rgb); }
/* one Verlet update of (x, v) under the caller's acceleration field acc = -k x (the harmonic
 * oscillator of advanced-physics.test.ts:199-218), through the march's own two update lines */
void orc_hook_glsl_verlet_oscillator(float *x, float *v, float k, float dt, int steps) {
    v3 p = {*x, 0.0f, 0.0f}, w = {*v, 0.0f, 0.0f};
    for (int i = 0; i < steps; i++) {
        v3 a0 = {-k * p.x, 0.0f, 0.0f};
        p = gl_verlet_position(p, w, a0, dt);
        v3 a1 = {-k * p.x, 0.0f, 0.0f};
        w = gl_verlet_velocity(w, a0, a1, dt);
    }
    *x = p.x;
    *v = w.x;
}
