// shader_kernels.hpp -- f32 march loops of the reference's two GPU shaders as HIP
// kernels (compiled in the -ffp-contract=off unit so operation order is the shader's).
//
//   wgsl_symplectic_kernel : src/shaders/compute.wgsl.ts:28-258  (Kerr-Schild Hamiltonian,
//       implicit midpoint with 2 fixed-point sweeps, h = clamp((r-r+)*0.15, 0.05, 1),
//       thin-disk crossing g-factor shading)                                  [SURVEY a18]
//   glsl_verlet_kernel     : src/shaders/blackhole/fragment.glsl.ts:40-221 with
//       chunks/metric.ts:13-149 (Cartesian velocity-Verlet on the Kerr pseudo-potential,
//       ZAMO twist) [a16] and chunks/disk.ts:16-115 + chunks/blackbody.ts:9-34 (disk
//       Doppler / beaming / temperature block) [a17]
//
// Not reproducible in the reference and therefore fixed here: WGSL star hash
// (omitted), GLSL blue-noise dither (0), GLSL turbulence noise() fetches (caller constant).
// One thread per pixel, registers only; a wave is one 8x8 pixel block (compute.wgsl.ts:147).
#pragma once

#include "frame_kernels.hpp"

namespace {

struct F3 {
    float x, y, z;
};
__device__ __forceinline__ float clampf_d(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
__device__ __forceinline__ float dot_f3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross_f3(F3 a, F3 b) {
    return F3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ F3 scale_f3(F3 a, float s) { return F3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ F3 add_f3(F3 a, F3 b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ float length_f3(F3 a) { return sqrtf(dot_f3(a, a)); }
__device__ __forceinline__ F3 normalize_f3(F3 a) {
    const float l = length_f3(a);
    return F3{a.x / l, a.y / l, a.z / l};
}
__device__ __forceinline__ float smoothstep_d(float e0, float e1, float x) {
    const float t = clampf_d((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
__device__ __forceinline__ float sign_d(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// block-level sum of the per-thread step counts into one atomic
__device__ __forceinline__ void add_steps(unsigned long long *total, uint32_t steps) {
    __shared__ unsigned long long s_w[16];
    unsigned long long v = wave_sum_u64(steps);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) s_w[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (uint32_t w = 0; w < (blockDim.x + 63u) / 64u; ++w) t += s_w[w];
        if (t) atomicAdd(total, t);
    }
}

// ===========================================================================
// WGSL compute kernel
// ===========================================================================
struct Ray32 {
    float t, r, th, ph, pt, pr, pth, pph;
};
struct D32 {
    float dt, dr, dth, dph, dpr, dpth;
};

// compute.wgsl.ts:42-120 (dp_t = dp_phi = 0)
__device__ __forceinline__ D32 wgsl_derivs(const Ray32 &s, float M, float spin) {
    const float a = spin * M;
    const float r = s.r, theta = s.th;
    const float r2 = r * r, a2 = a * a;
    const float sint = sinf(theta), cost = cosf(theta);
    const float sin2 = fmaxf(sint * sint, 1e-12f);
    const float cos2 = 1.0f - sin2;
    const float sigma = r2 + a2 * cos2;
    const float sigma2 = sigma * sigma;
    const float delta = r2 - 2.0f * M * r + a2;

    const float g_tt = -(1.0f + 2.0f * M * r / sigma);
    const float g_tr = 2.0f * M * r / sigma;
    const float g_rr = delta / sigma;
    const float g_thth = 1.0f / sigma;
    const float g_phph = 1.0f / (sigma * sin2);
    const float g_rph = a / sigma;
    D32 d;
    d.dt = g_tt * s.pt + g_tr * s.pr;
    d.dr = g_tr * s.pt + g_rr * s.pr + g_rph * s.pph;
    d.dth = g_thth * s.pth;
    d.dph = g_rph * s.pr + g_phph * s.pph;

    const float dsigma_dr = 2.0f * r;
    const float dsigma_dth = -2.0f * a2 * sint * cost;
    const float ddelta_dr = 2.0f * r - 2.0f * M;
    const float dg_tt_dr = -(2.0f * M * (sigma - r * dsigma_dr)) / sigma2;
    const float dg_tt_dth = (2.0f * M * r * dsigma_dth) / sigma2;
    const float dg_tr_dr = -dg_tt_dr;
    const float dg_tr_dth = -dg_tt_dth;
    const float dg_rr_dr = (ddelta_dr * sigma - delta * dsigma_dr) / sigma2;
    const float dg_rr_dth = -(delta * dsigma_dth) / sigma2;
    const float dg_thth_dr = -dsigma_dr / sigma2;
    const float dg_thth_dth = -dsigma_dth / sigma2;
    const float dg_phph_dr = -dsigma_dr / (sigma2 * sin2);
    const float dg_phph_dth =
        -(dsigma_dth * sin2 + sigma * sinf(2.0f * theta)) / (sigma2 * sin2 * sin2);
    const float dg_rph_dr = -(a * dsigma_dr) / sigma2;
    const float dg_rph_dth = -(a * dsigma_dth) / sigma2;

    const float dh_dr = 0.5f * (dg_tt_dr * s.pt * s.pt + dg_rr_dr * s.pr * s.pr +
                                dg_thth_dr * s.pth * s.pth + dg_phph_dr * s.pph * s.pph +
                                2.0f * dg_tr_dr * s.pt * s.pr + 2.0f * dg_rph_dr * s.pr * s.pph);
    const float dh_dth = 0.5f * (dg_tt_dth * s.pt * s.pt + dg_rr_dth * s.pr * s.pr +
                                 dg_thth_dth * s.pth * s.pth + dg_phph_dth * s.pph * s.pph +
                                 2.0f * dg_tr_dth * s.pt * s.pr + 2.0f * dg_rph_dth * s.pr * s.pph);
    d.dpr = -dh_dr;
    d.dpth = -dh_dth;
    return d;
}

// compute.wgsl.ts:122-133
__device__ __forceinline__ Ray32 wgsl_symplectic(const Ray32 &s, float h, float M, float spin) {
    Ray32 mid = s;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const D32 d = wgsl_derivs(mid, M, spin);
        mid.t = (s.t + (s.t + d.dt * h)) * 0.5f;
        mid.r = (s.r + (s.r + d.dr * h)) * 0.5f;
        mid.th = (s.th + (s.th + d.dth * h)) * 0.5f;
        mid.ph = (s.ph + (s.ph + d.dph * h)) * 0.5f;
        mid.pt = (s.pt + (s.pt + 0.0f * h)) * 0.5f;
        mid.pr = (s.pr + (s.pr + d.dpr * h)) * 0.5f;
        mid.pth = (s.pth + (s.pth + d.dpth * h)) * 0.5f;
        mid.pph = (s.pph + (s.pph + 0.0f * h)) * 0.5f;
    }
    const D32 f = wgsl_derivs(mid, M, spin);
    Ray32 o;
    o.t = s.t + f.dt * h;
    o.r = s.r + f.dr * h;
    o.th = s.th + f.dth * h;
    o.ph = s.ph + f.dph * h;
    o.pt = s.pt + 0.0f * h;
    o.pr = s.pr + f.dpr * h;
    o.pth = s.pth + f.dpth * h;
    o.pph = s.pph + 0.0f * h;
    return o;
}

__device__ __forceinline__ void m4v4_f(const float *m, float x, float y, float z, float w, float o[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = m[0 + r] * x + m[4 + r] * y + m[8 + r] * z + m[12 + r] * w;
}

// compute.wgsl.ts:147-258
__global__ __launch_bounds__(kBlock) void wgsl_symplectic_kernel(FrameGeom G, WgslParams P,
                                                                 float4 *__restrict__ out_rgba,
                                                                 uint32_t *__restrict__ out_steps,
                                                                 unsigned long long *total_steps,
                                                                 uint32_t n_slots) {
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
    uint32_t X = 0, Y = 0, oi = 0;
    const bool valid = slot < n_slots && slot_to_pixel(G, slot, X, Y, oi);
    uint32_t steps = 0;
    if (valid) {
        const float PI = 3.14159265f;
        const float fw = (float)G.width, fh = (float)G.height;
        const float jx = P.jitter[0] / fw, jy = P.jitter[1] / fh;
        const float uvx = (float)X / fw, uvy = (float)Y / fh;
        const float ndcx = (uvx + jx) * 2.0f - 1.0f;
        const float ndcy = (uvy + jy) * 2.0f - 1.0f;
        float vt[4];
        m4v4_f(P.inv_proj, ndcx, -ndcy, 1.0f, 1.0f, vt);
        F3 vd = normalize_f3(F3{vt[0] / vt[3], vt[1] / vt[3], vt[2] / vt[3]});
        float w4[4];
        m4v4_f(P.inv_view, vd.x, vd.y, vd.z, 0.0f, w4);
        const F3 wd = normalize_f3(F3{w4[0], w4[1], w4[2]});

        const F3 cam{P.position[0], P.position[1], P.position[2]};
        const float r0 = length_f3(cam);
        const float theta0 = acosf(clampf_d(cam.y / r0, -1.0f, 1.0f));
        const float phi0 = atan2f(cam.z, cam.x);
        const float st = sinf(theta0), ct = cosf(theta0), sp = sinf(phi0), cp = cosf(phi0);
        const float pr_far = dot_f3(wd, F3{st * cp, ct, st * sp});
        const float pth_far = dot_f3(wd, F3{ct * cp, -st, ct * sp}) / r0;
        const float safe_st = fmaxf(st, 1e-4f);
        const float pph_far = dot_f3(wd, F3{-sp, 0.0f, cp}) / (r0 * safe_st);

        Ray32 s;
        s.t = 0.0f;
        s.r = r0;
        s.th = theta0;
        s.ph = phi0;
        s.pt = -1.0f;
        s.pr = pr_far;
        s.pth = pth_far * r0 * r0;
        s.pph = pph_far * r0 * r0 * st * st;

        const float M = P.mass;
        const float a = P.spin * M;
        // compute.wgsl.ts:28-40
        const float disc = M * M - a * a;
        const float rh = disc < 0.0f ? M : M + sqrtf(disc);
        const float absS = fabsf(clampf_d(a / M, -0.999f, 0.999f));
        const float z1 = 1.0f + powf(1.0f - absS * absS, 1.0f / 3.0f) *
                                    (powf(1.0f + absS, 1.0f / 3.0f) + powf(1.0f - absS, 1.0f / 3.0f));
        const float z2 = sqrtf(3.0f * absS * absS + z1 * z1);
        const float isco = M * (3.0f + z2 - sqrtf((3.0f - z1) * (3.0f + z1 + 2.0f * z2)));

        float col[3] = {0.0f, 0.0f, 0.0f};
        float alpha = 0.0f;
        for (int i = 0; i < P.max_steps; ++i) {
            const float r = s.r;
            if (r < rh * 1.001f) break;
            if (r > 100.0f) break;
            const float prev_theta = s.th;
            const float h = clampf_d((r - rh) * 0.15f, 0.05f, 1.0f);
            s = wgsl_symplectic(s, h, M, P.spin);
            ++steps;
            const float curr_theta = s.th;
            if ((prev_theta - PI * 0.5f) * (curr_theta - PI * 0.5f) <= 0.0f && r > isco && r < 30.0f) {
                const float Omega = 1.0f / (powf(r, 1.5f) + a);
                const float u_t =
                    1.0f / sqrtf(fmaxf(1.0f - 2.0f * M / r - Omega * Omega * (r * r + a * a), 1e-4f));
                const float u_phi = Omega * u_t;
                const float g_factor = -s.pt / fmaxf(-(u_t * s.pt + u_phi * s.pph), 1e-4f);
                const float artistic_T = (1.0f / powf(fmaxf(r / isco, 1.0f), 0.75f)) * g_factor;
                const float base[3] = {1.0f, 0.5f, 0.1f}, blue[3] = {0.5f, 0.7f, 1.0f},
                            red[3] = {1.0f, 0.2f, 0.0f};
                const float bs = fmaxf(g_factor - 1.0f, 0.0f), rs = fmaxf(1.0f - g_factor, 0.0f) * 0.5f;
                const float target_opacity = 0.6f * artistic_T;
                const float g4 = powf(g_factor, 4.0f);
                const float mri_shear = powf(r, -1.5f);
                const float mri_sat = 1.0f + 0.0001f * sinf(r * 100.0f * mri_shear);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float target = (base[c] + blue[c] * bs - red[c] * rs) * artistic_T * 4.0f;
                    const float I_em = target * target_opacity / fmaxf(g4, 1e-5f);
                    const float j_nu = I_em * mri_sat;
                    const float I_obs = g4 * j_nu;
                    col[c] += I_obs * (1.0f - alpha);
                }
                alpha += target_opacity * mri_sat;
            }
            if (alpha > 0.99f) break;
        }
        if (out_rgba) out_rgba[oi] = make_float4(col[0], col[1], col[2], 1.0f);
        if (out_steps) out_steps[oi] = steps;
    }
    add_steps(total_steps, steps);
}

// ===========================================================================
// GLSL fragment march
// ===========================================================================
// chunks/metric.ts:96-149
__device__ __forceinline__ F3 glsl_kerr_accel(F3 p, F3 v, float M, float a, float &omega) {
    const float a2 = a * a;
    const float rho2 = dot_f3(p, p);
    const float diff = rho2 - a2;
    const float disc = diff * diff + 4.0f * a2 * p.y * p.y;
    const float r2 = 0.5f * (diff + sqrtf(fmaxf(0.0f, disc)));
    const float r_k = sqrtf(fmaxf(1e-8f, r2));
    const float sigma = r2 + a2 * (p.y * p.y / fmaxf(1e-8f, r2));
    const F3 L = cross_f3(p, v);
    const float Ly = L.y;
    const float Ly_eff = Ly - a;
    const float L2_eff = Ly_eff * Ly_eff + (dot_f3(L, L) - Ly * Ly);
    const float r_inv = 1.0f / r_k;
    const float r2_inv = r_inv * r_inv;
    const float r4_inv = r2_inv * r2_inv;
    const float sigma_ratio = r2 / fmaxf(1e-8f, sigma);
    const F3 n = normalize_f3(p);
    const F3 r_hat{-n.x, -n.y, -n.z};
    F3 acc = scale_f3(r_hat, M * r2_inv * sigma_ratio + 3.0f * M * fmaxf(0.0f, L2_eff) * r4_inv * sigma_ratio);
    const float r3_p_a2r = r_k * r2 + a2 * r_k;
    const float drag = 2.0f * M * a / fmaxf(1e-8f, r3_p_a2r);
    acc = add_f3(acc, scale_f3(cross_f3(F3{0.0f, 1.0f, 0.0f}, v), drag));
    omega = 2.0f * M * a / fmaxf(1e-8f, r3_p_a2r);
    return acc;
}

// chunks/common.ts:44-47 : `v.ab *= rot(ang)`
__device__ __forceinline__ void glsl_rot(float ang, float &x, float &y) {
    const float s = sinf(ang), c = cosf(ang);
    const float nx = x * c + y * (-s);
    const float ny = x * s + y * c;
    x = nx;
    y = ny;
}

// chunks/blackbody.ts:9-34
__device__ __forceinline__ void glsl_blackbody(float temp, float rgb[3]) {
    const float t = fmaxf(temp, 1.0f) / 100.0f;
    float r, g, b;
    if (t <= 66.0f) {
        r = 255.0f;
        g = 99.4708025861f * logf(t) - 161.1195681661f;
        b = (t <= 19.0f) ? 0.0f : 138.5177312231f * logf(t - 10.0f) - 305.0447927307f;
    } else {
        r = 329.698727446f * powf(t - 60.0f, -0.1332047592f);
        g = 288.1221695283f * powf(t - 60.0f, -0.0755148492f);
        b = 255.0f;
    }
    rgb[0] = powf(fmaxf(r / 255.0f, 0.0f), 2.2f);
    rgb[1] = powf(fmaxf(g / 255.0f, 0.0f), 2.2f);
    rgb[2] = powf(fmaxf(b / 255.0f, 0.0f), 2.2f);
}

// chunks/disk.ts:16-115
__device__ __forceinline__ void glsl_sample_disk(const GlslParams &U, F3 p, F3 p_prev, F3 v,
                                                 float isco, float M, float a, float dt,
                                                 float col[3], float &alpha) {
    const bool crossed = (p_prev.y * p.y < 0.0f);
    F3 sp = p;
    if (crossed) {
        const float t = fabsf(p_prev.y) / fmaxf(0.0001f, fabsf(p_prev.y) + fabsf(p.y));
        sp.x = p_prev.x * (1.0f - t) + p.x * t;
        sp.y = p_prev.y * (1.0f - t) + p.y * t;
        sp.z = p_prev.z * (1.0f - t) + p.z * t;
    }
    const float sampleR = length_f3(sp);
    const float effH = fminf(U.disk_scale_height, 0.45f);
    const float diskHeight = sampleR * effH;
    const float diskInner = isco;
    const float diskOuter = fmaxf(M * U.disk_size, diskInner * 1.1f);
    if (!((fabsf(sp.y) < diskHeight || crossed) && sampleR > diskInner && sampleR < diskOuter)) return;
    const float heightFalloff = expf(-fabsf(sp.y) / fmaxf(0.001f, (sampleR * effH) * 0.25f));
    const float radialFalloff = smoothstep_d(diskOuter, diskInner, sampleR);
    const float baseDensity = U.turbulence * heightFalloff * radialFalloff;
    if (!(baseDensity > 0.001f)) return;

    const float r2 = sampleR * sampleR;
    const float sqrt_M = sqrtf(M);
    const float signSpin = sign_d(U.spin + 1e-8f);
    const float Omega = (signSpin * sqrt_M) / (sampleR * sqrtf(sampleR) + a * sqrt_M);
    const float g_tt = -(1.0f - 2.0f * M / sampleR);
    const float g_tphi = -2.0f * M * a / sampleR;
    const float g_phiphi = r2 + a * a + 2.0f * M * a * a / sampleR;
    const float u_t_sq = -(g_tt + 2.0f * Omega * g_tphi + Omega * Omega * g_phiphi);
    const float u_t = 1.0f / sqrtf(fmaxf(1e-6f, u_t_sq));
    const float L_photon = p.z * v.x - p.x * v.z;
    const float delta = 1.0f / fmaxf(0.01f, u_t * (1.0f - Omega * L_photon));
    const float beaming = fmaxf(0.01f, powf(delta, 3.5f));
    const float isco_r = clampf_d(isco / sampleR, 0.0f, 1.0f);
    const float nt_factor = fmaxf(0.0f, 1.0f - sqrtf(isco_r));
    const float grad = powf(isco_r, 0.75f) * powf(nt_factor, 0.25f);
    const float temperature = U.disk_temp * grad * delta;
    float bb[3];
    glsl_blackbody(temperature, bb);
    const float density = baseDensity * U.disk_density * 0.12f * dt;
#pragma unroll
    for (int c = 0; c < 3; ++c) col[c] += bb[c] * beaming * density * (1.0f - alpha);
    alpha += density;
}

__device__ __forceinline__ float aces_d(float c) {
    return clampf_d((c * (2.51f * c + 0.03f)) / (c * (2.43f * c + 0.59f) + 0.14f), 0.0f, 1.0f);
}

// fragment.glsl.ts:40-221 (+276, 327-333), fallback camera
__global__ __launch_bounds__(kBlock) void glsl_verlet_kernel(FrameGeom G, GlslParams U,
                                                             float4 *__restrict__ out_rgba,
                                                             uint32_t *__restrict__ out_steps,
                                                             unsigned long long *total_steps,
                                                             uint32_t n_slots) {
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
    uint32_t X = 0, Y = 0, oi = 0;
    const bool valid = slot < n_slots && slot_to_pixel(G, slot, X, Y, oi);
    uint32_t steps = 0;
    if (valid) {
        const float PI = 3.14159265359f;
        const float resx = (float)G.width, resy = (float)G.height;
        const float minRes = fminf(resx, resy);
        const float fcx = (float)X + 0.5f, fcy = (float)(G.height - 1u - Y) + 0.5f;
        const float uvx = (fcx - 0.5f * resx) / minRes, uvy = (fcy - 0.5f * resy) / minRes;
        F3 ro{0.0f, 0.0f, -U.zoom};
        F3 rd = normalize_f3(F3{uvx, uvy, 1.5f});
        const float ax = (U.mouse[1] - 0.5f) * PI, ay = (U.mouse[0] - 0.5f) * PI * 2.0f;
        glsl_rot(ax, ro.y, ro.z);
        glsl_rot(ax, rd.y, rd.z);
        glsl_rot(ay, ro.x, ro.z);
        glsl_rot(ay, rd.x, rd.z);

        const float M = U.mass;
        const float a = U.spin * M;
        const float rh = M + sqrtf(fmaxf(0.0f, M * M - a * a)); // metric.ts:13-15
        // metric.ts:32-37
        const float a_star = clampf_d(a / M, -0.9999f, 0.9999f);
        const float rph = 2.0f * M * (1.0f + cosf((2.0f / 3.0f) * acosf(clampf_d(-a_star, -1.0f, 1.0f))));
        // metric.ts:18-29
        const float absS = fabsf(clampf_d(a / M, -0.9999f, 0.9999f));
        const float z1 = 1.0f + powf(1.0f - absS * absS, 1.0f / 3.0f) *
                                    (powf(1.0f + absS, 1.0f / 3.0f) + powf(1.0f - absS, 1.0f / 3.0f));
        const float z2 = sqrtf(3.0f * absS * absS + z1 * z1);
        float sgnA = sign_d(a);
        if (sgnA == 0.0f) sgnA = 1.0f;
        const float isco = M * (3.0f + z2 - sgnA * sqrtf((3.0f - z1) * (3.0f + z1 + 2.0f * z2)));

        F3 p = ro, v = rd;
        if (length_f3(ro) < rh * 1.5f) {
            ro = scale_f3(scale_f3(normalize_f3(ro), rh), 1.5f);
            p = ro;
        }
        float col[3] = {0.0f, 0.0f, 0.0f};
        float alpha = 0.0f;
        const int maxSteps = (int)fminf((float)U.max_ray_steps, 500.0f);
        F3 p_prev = p;
        for (int i = 0; i < maxSteps; ++i) {
            p_prev = p;
            const float r = length_f3(p);
            if (r < rh * 1.15f) break;
            if (r > 10000.0f) break;
            const float distFactor = 1.0f + r * 0.05f;
            float dt = clampf_d((r - rh) * 0.1f * distFactor, 0.01f, 1.2f * distFactor);
            if (r > 30.0f) {
                const float farBoost = (r - 30.0f) * 0.08f;
                dt = fmaxf(dt, 0.01f + farBoost);
                dt = fminf(dt, 1.2f * 2.5f);
            }
            const float sphereProx = fabsf(r - rph);
            dt = fminf(dt, 0.01f + sphereProx * 0.15f);
            const float hRefinement = smoothstep_d(0.2f, 0.0f, fabsf(p.y));
            const float cdt = dt * (1.0f - hRefinement * 0.7f);

            float omega, om2;
            const F3 accel = scale_f3(glsl_kerr_accel(p, v, M, a, omega), U.lensing_strength);
            glsl_rot(omega * cdt, v.x, v.z);
            p = add_f3(p, add_f3(scale_f3(v, cdt), scale_f3(scale_f3(scale_f3(accel, 0.5f), cdt), cdt)));
            if (alpha < 0.95f) {
                const F3 accel_new = scale_f3(glsl_kerr_accel(p, v, M, a, om2), U.lensing_strength);
                v = add_f3(v, scale_f3(scale_f3(add_f3(accel, accel_new), 0.5f), cdt));
            }
            v = normalize_f3(v);
            ++steps;
            glsl_sample_disk(U, p, p_prev, v, isco, M, a, cdt, col, alpha);
            if (alpha > 0.99f) break;
        }
        float o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = U.tone_map ? powf(fmaxf(aces_d(col[c]), 0.0f), 0.4545f) : col[c];
        if (out_rgba) out_rgba[oi] = make_float4(o[0], o[1], o[2], 1.0f);
        if (out_steps) out_steps[oi] = steps;
    }
    add_steps(total_steps, steps);
}

} // namespace
