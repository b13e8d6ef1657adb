// shader_kernels.hpp -- f32 march loops of the reference's two GPU shaders as HIP
// kernels (compiled in the -ffp-contract=off unit so operation order is the shader's).
//
//   wgsl_symplectic_kernel : src/shaders/compute.wgsl.ts:28-258  (Kerr-Schild Hamiltonian,
//       implicit midpoint with 2 fixed-point sweeps, h = clamp((r-r+)*0.15, 0.05, 1),
//       thin-disk crossing g-factor shading)                                  [SURVEY a18]
//   glsl_fragment_kernel   : src/shaders/blackhole/fragment.glsl.ts:40-334, the whole main():
//       chunks/metric.ts:13-149 (Cartesian velocity-Verlet on the Kerr pseudo-potential,
//       ZAMO twist) [a16], chunks/disk.ts:16-115 + chunks/blackbody.ts:9-34 (disk Doppler /
//       beaming / temperature block) [a17], and the compositing of SURVEY 8f-3: disk
//       turbulence + jets (chunks/disk.ts:43-55, 117-155), starfield (chunks/background.ts),
//       texture noise (chunks/noise.ts), photon-ring / ergosphere glow, redshift overlay,
//       Kerr-shadow guide, ACES (fragment.glsl.ts:197-333).  ShaderManager's #defines
//       (manager.ts:61-82) are GlslParams::features bits.
//
// WGSL star hash (compute.wgsl.ts:201-204): restated with sinf; fract(sin * 43758.5) turns the
// last ulp of sin into 5e-3, so a few star pixels can differ between libms (WgslParams::stars).
// Not reproducible in the reference and therefore fixed here: the two GLSL noise textures are Math.random() upstream -- here they are
// engine-owned seeded 256x256 byte planes (grv_set_glsl_noise), sampled with f32 weights.
// One thread per pixel, registers only; a wave is one 8x8 pixel block (compute.wgsl.ts:147).
#pragma once

#include "frame_kernels.hpp"
#include "shader_common.hpp"

namespace {

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
    const float sint = sh_sinf(theta), cost = sh_cosf(theta);
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
        -(dsigma_dth * sin2 + sigma * sh_sinf(2.0f * theta)) / (sigma2 * sin2 * sin2);
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
__global__ __launch_bounds__(kMarchBlock) void wgsl_symplectic_kernel(FrameGeom G, WgslParams P,
                                                                 float4 *__restrict__ out_rgba,
                                                                 uint32_t *__restrict__ out_steps,
                                                                 unsigned long long *total_steps,
                                                                 uint32_t n_slots) {
    const uint32_t slot = blockIdx.x * kMarchBlock + threadIdx.x;
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
        const float theta0 = sh_acosf(clampf_d(cam.y / r0, -1.0f, 1.0f));
        const float phi0 = sh_atan2f(cam.z, cam.x);
        const float st = sh_sinf(theta0), ct = sh_cosf(theta0), sp = sh_sinf(phi0), cp = sh_cosf(phi0);
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
        const float z1 = 1.0f + sh_powf(1.0f - absS * absS, 1.0f / 3.0f) *
                                    (sh_powf(1.0f + absS, 1.0f / 3.0f) + sh_powf(1.0f - absS, 1.0f / 3.0f));
        const float z2 = sqrtf(3.0f * absS * absS + z1 * z1);
        const float isco = M * (3.0f + z2 - sqrtf((3.0f - z1) * (3.0f + z1 + 2.0f * z2)));

        float col[3] = {0.0f, 0.0f, 0.0f};
        float alpha = 0.0f;
        // the shader's operations in their order, in a loop with ONE exit (control flow only: the three
        // places the shader leaves from -- horizon and far tests at the top, the opaque test at the
        // bottom -- are tested together at the top of the next iteration; the escape branch's star hash
        // runs after the loop on the rays that left through r > 100)
        int i = 0;
        bool opaque = false, below = false, far = false;
        for (;;) {
            const float r = s.r;
            below = r < rh * 1.001f;
            far = r > 100.0f;
            if (!(i < P.max_steps) || opaque || below || far) break;
            const float prev_theta = s.th;
            const float h = clampf_d((r - rh) * 0.15f, 0.05f, 1.0f);
            s = wgsl_symplectic(s, h, M, P.spin);
            ++i;
            const float curr_theta = s.th;
            if ((prev_theta - PI * 0.5f) * (curr_theta - PI * 0.5f) <= 0.0f && r > isco && r < 30.0f) {
                const float Omega = 1.0f / (sh_powf(r, 1.5f) + a);
                const float u_t =
                    1.0f / sqrtf(fmaxf(1.0f - 2.0f * M / r - Omega * Omega * (r * r + a * a), 1e-4f));
                const float u_phi = Omega * u_t;
                const float g_factor = -s.pt / fmaxf(-(u_t * s.pt + u_phi * s.pph), 1e-4f);
                const float artistic_T = (1.0f / sh_powf(fmaxf(r / isco, 1.0f), 0.75f)) * g_factor;
                const float base[3] = {1.0f, 0.5f, 0.1f}, blue[3] = {0.5f, 0.7f, 1.0f},
                            red[3] = {1.0f, 0.2f, 0.0f};
                const float bs = fmaxf(g_factor - 1.0f, 0.0f), rs = fmaxf(1.0f - g_factor, 0.0f) * 0.5f;
                const float target_opacity = 0.6f * artistic_T;
                const float g4 = sh_powf(g_factor, 4.0f);
                const float mri_shear = sh_powf(r, -1.5f);
                const float mri_sat = 1.0f + 0.0001f * sh_sinf(r * 100.0f * mri_shear);
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
            opaque = alpha > 0.99f;
        }
        steps = (uint32_t)i;
        if (P.stars && far && !below && !opaque && i < P.max_steps) { // compute.wgsl.ts:199-206
            const float r = s.r;
            const F3 vdir = normalize_f3(F3{s.pr, s.pth / r, s.pph / (r * safe_st)});
            const float sn = sh_sinf(vdir.x * 12.9898f + vdir.y * 78.233f + vdir.z * 45.164f) * 43758.5453f;
            if (sn - floorf(sn) > 0.999f)
                for (int c = 0; c < 3; ++c) col[c] += 1.0f * (1.0f - alpha);
        }
        if (out_rgba) out_rgba[oi] = make_float4(col[0], col[1], col[2], 1.0f);
        if (out_steps) out_steps[oi] = steps;
    }
    add_steps(total_steps, steps, blockIdx.x);
}

} // namespace

#include "glsl_fragment.hpp"
