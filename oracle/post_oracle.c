/* post_oracle.c -- see post_oracle.h.  TEST INFRASTRUCTURE ONLY.  Shader evaluation order, f32. */
#include "post_oracle.h"
#include "ref_libm.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
static float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

/* binary32 -> binary16 -> binary32, round to nearest even, overflow to inf */
float orc_round_to_half(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint32_t sign = u & 0x80000000u;
    uint32_t a = u & 0x7FFFFFFFu;
    float r;
    if (a >= 0x7F800000u) return x;                      /* inf / nan */
    if (a >= 0x477FF000u) {                              /* >= 65520 rounds to inf */
        a = 0x7F800000u;
    } else if (a < 0x38800000u) {                        /* below 2^-14: half subnormal, step 2^-24 */
        float f;
        memcpy(&f, &a, 4);
        f = f * 16777216.0f;                             /* exact scaling by 2^24 */
        f = nearbyintf(f);                               /* RNE in the default rounding mode */
        f = f * (1.0f / 16777216.0f);
        memcpy(&a, &f, 4);
    } else {                                             /* normal: keep 10 mantissa bits */
        const uint32_t lsb = (a >> 13) & 1u;
        a += 0x00000FFFu + lsb;
        a &= 0xFFFFE000u;
    }
    a |= sign;
    memcpy(&r, &a, 4);
    return r;
}

static float store(float v, int half) { return half ? orc_round_to_half(v) : v; }

/* texture(tex, uv) with LINEAR + CLAMP_TO_EDGE on a w x h RGBA f32 image */
static void sample_linear(const float *tex, uint32_t w, uint32_t h, float u, float v, float out[4]) {
    const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float a = x - fx, b = y - fy;
    const int i0 = clampi((int)fx, 0, (int)w - 1), i1 = clampi((int)fx + 1, 0, (int)w - 1);
    const int j0 = clampi((int)fy, 0, (int)h - 1), j1 = clampi((int)fy + 1, 0, (int)h - 1);
    const float *t00 = tex + 4 * ((size_t)j0 * w + i0), *t10 = tex + 4 * ((size_t)j0 * w + i1);
    const float *t01 = tex + 4 * ((size_t)j1 * w + i0), *t11 = tex + 4 * ((size_t)j1 * w + i1);
    for (int c = 0; c < 4; c++)
        out[c] = (1.0f - a) * (1.0f - b) * t00[c] + a * (1.0f - b) * t10[c] + (1.0f - a) * b * t01[c] +
                 a * b * t11[c];
}

static void rgb_to_ycocg(const float rgb[3], float o[3]) {
    o[0] = rgb[0] * 0.25f + rgb[1] * 0.5f + rgb[2] * 0.25f;
    o[1] = rgb[0] * 0.5f + rgb[1] * 0.0f + rgb[2] * -0.5f;
    o[2] = rgb[0] * -0.25f + rgb[1] * 0.5f + rgb[2] * -0.25f;
}
static void ycocg_to_rgb(const float y[3], float o[3]) {
    o[0] = y[0] + y[1] - y[2];
    o[1] = y[0] + y[2];
    o[2] = y[0] - y[1] - y[2];
}

void orc_taa_resolve(uint32_t w, uint32_t h, const float *current, const float *history,
                     float blend_factor, int camera_moving, int half_storage, float *out) {
    const float tx = 1.0f / (float)w, ty = 1.0f / (float)h; /* scale (1,1) / u_resolution */
#pragma omp parallel for schedule(static)
    for (long long k = 0; k < (long long)w * h; k++) {
        const uint32_t px = (uint32_t)(k % w), py = (uint32_t)(k / w);
        const float u = ((float)px + 0.5f) / (float)w, v = ((float)py + 0.5f) / (float)h;
        float cur[4], s4[4], s[3], m1[3] = {0, 0, 0}, m2[3] = {0, 0, 0};
        sample_linear(current, w, h, u, v, cur);
        for (int y = -1; y <= 1; y++)
            for (int x = -1; x <= 1; x++) {
                sample_linear(current, w, h, u + (float)x * tx, v + (float)y * ty, s4);
                rgb_to_ycocg(s4, s);
                for (int c = 0; c < 3; c++) {
                    m1[c] += s[c];
                    m2[c] += s[c] * s[c];
                }
            }
        float mean[3], sd[3], hist4[4], hy[3], cy[3], res[3], rgb[3];
        for (int c = 0; c < 3; c++) {
            mean[c] = m1[c] / 9.0f;
            sd[c] = sqrtf(fmaxf(m2[c] / 9.0f - mean[c] * mean[c], 0.0f));
        }
        sample_linear(history, w, h, u, v, hist4);
        rgb_to_ycocg(hist4, hy);
        for (int c = 0; c < 3; c++) hy[c] = clampf(hy[c], mean[c] - 1.5f * sd[c], mean[c] + 1.5f * sd[c]);
        const float varianceWeight = 1.0f - clampf(sd[0] * 4.0f, 0.0f, 0.55f);
        const float alpha = camera_moving ? 0.0f : blend_factor * varianceWeight;
        rgb_to_ycocg(cur, cy);
        for (int c = 0; c < 3; c++) res[c] = mixf(cy[c], hy[c], alpha);
        ycocg_to_rgb(res, rgb);
        float *o = out + 4 * k;
        for (int c = 0; c < 3; c++) o[c] = store(rgb[c], half_storage);
        o[3] = 1.0f;
    }
}

static void m4v4(const float *m, const float v[4], float o[4]) { /* column-major m * v */
    for (int r = 0; r < 4; r++) o[r] = m[0 + r] * v[0] + m[4 + r] * v[1] + m[8 + r] * v[2] + m[12 + r] * v[3];
}

void orc_ataa_resolve(uint32_t w, uint32_t h, const orc_ataa_camera *cam, const float *current,
                      const float *history, int half_storage, float *out) {
#pragma omp parallel for schedule(static)
    for (long long k = 0; k < (long long)w * h; k++) {
        const int px = (int)(k % w), py = (int)(k / w);
        const float u = ((float)px + 0.5f) / (float)w, v = ((float)py + 0.5f) / (float)h;
        float m1[3] = {0, 0, 0}, m2[3] = {0, 0, 0}, center[3], s[3];
        rgb_to_ycocg(current + 4 * ((size_t)py * w + px), center);
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                const int sx = clampi(px + dx, 0, (int)w - 1), sy = clampi(py + dy, 0, (int)h - 1);
                rgb_to_ycocg(current + 4 * ((size_t)sy * w + sx), s);
                for (int c = 0; c < 3; c++) {
                    m1[c] += s[c];
                    m2[c] += s[c] * s[c];
                }
            }
        float mean[3], sd[3];
        for (int c = 0; c < 3; c++) {
            mean[c] = m1[c] / 9.0f;
            sd[c] = sqrtf(fmaxf(m2[c] / 9.0f - mean[c] * mean[c], 0.0f));
        }
        const float ndcx = u * 2.0f - 1.0f, ndcy = v * 2.0f - 1.0f;
        const float clip[4] = {ndcx, -ndcy, 1.0f, 1.0f};
        float vt[4], wd[4];
        m4v4(cam->inv_proj, clip, vt);
        float vx = vt[0] / vt[3], vy = vt[1] / vt[3], vz = vt[2] / vt[3];
        const float len = sqrtf(vx * vx + vy * vy + vz * vz);
        vx /= len;
        vy /= len;
        vz /= len;
        const float vd[4] = {vx, vy, vz, 0.0f};
        m4v4(cam->inv_view, vd, wd);
        const float depth = 12.0f;
        const float wp[4] = {cam->position[0] + wd[0] * depth, cam->position[1] + wd[1] * depth,
                             cam->position[2] + wd[2] * depth, 1.0f};
        float pc[4];
        m4v4(cam->prev_view_proj, wp, pc);
        const float pu = (pc[0] / pc[3]) * 0.5f + 0.5f, pv = (pc[1] / pc[3]) * -0.5f + 0.5f;
        float h4[4], hy[3], res[3], rgb[3];
        sample_linear(history, w, h, pu, pv, h4);
        rgb_to_ycocg(h4, hy);
        for (int c = 0; c < 3; c++) hy[c] = clampf(hy[c], mean[c] - 2.0f * sd[c], mean[c] + 2.0f * sd[c]);
        for (int c = 0; c < 3; c++) res[c] = mixf(center[c], hy[c], 0.92f);
        ycocg_to_rgb(res, rgb);
        float *o = out + 4 * k;
        for (int c = 0; c < 3; c++) o[c] = store(rgb[c], half_storage);
        o[3] = 1.0f;
    }
}

static float aces(float c) {
    return clampf((c * (2.51f * c + 0.03f)) / (c * (2.43f * c + 0.59f) + 0.14f), 0.0f, 1.0f);
}

static void blur_pass(const float *src, uint32_t sw, uint32_t sh, uint32_t bw, uint32_t bh, int vertical,
                      int half_storage, float *dst) {
    static const float wts[5] = {0.227027f, 0.1945946f, 0.1216216f, 0.054054f, 0.016216f};
    const float tx = 1.0f / (float)bw, ty = 1.0f / (float)bh; /* u_resolution = blur size */
#pragma omp parallel for schedule(static)
    for (long long k = 0; k < (long long)bw * bh; k++) {
        const uint32_t px = (uint32_t)(k % bw), py = (uint32_t)(k / bw);
        const float u = ((float)px + 0.5f) / (float)bw, v = ((float)py + 0.5f) / (float)bh;
        float c0[4], a[4], b[4], res[3];
        sample_linear(src, sw, sh, u, v, c0);
        for (int c = 0; c < 3; c++) res[c] = c0[c] * wts[0];
        for (int i = 1; i < 5; i++) {
            const float ox = (vertical ? 0.0f : 1.0f) * tx * (float)i, oy = (vertical ? 1.0f : 0.0f) * ty * (float)i;
            sample_linear(src, sw, sh, u + ox, v + oy, a);
            sample_linear(src, sw, sh, u - ox, v - oy, b);
            for (int c = 0; c < 3; c++) {
                res[c] += a[c] * wts[i];
                res[c] += b[c] * wts[i];
            }
        }
        float *o = dst + 4 * k;
        for (int c = 0; c < 3; c++) o[c] = store(res[c], half_storage);
        o[3] = 1.0f;
    }
}

void orc_bloom(uint32_t w, uint32_t h, const float *scene, float threshold, float intensity,
               int blur_passes, int half_storage, float *out) {
    const uint32_t hw = w / 2 ? w / 2 : 1, hh = h / 2 ? h / 2 : 1;
    const uint32_t bw = w / 4 ? w / 4 : 1, bh = h / 4 ? h / 4 : 1;
    float *bright = (float *)calloc((size_t)hw * hh * 4, sizeof(float));
    float *b1 = (float *)calloc((size_t)bw * bh * 4, sizeof(float));
    float *b2 = (float *)calloc((size_t)bw * bh * 4, sizeof(float));
#pragma omp parallel for schedule(static)
    for (long long k = 0; k < (long long)hw * hh; k++) { /* bright pass, bloom.glsl.ts:35-58 */
        const uint32_t px = (uint32_t)(k % hw), py = (uint32_t)(k / hw);
        float c[4];
        sample_linear(scene, w, h, ((float)px + 0.5f) / (float)hw, ((float)py + 0.5f) / (float)hh, c);
        const float lum = c[0] * 0.299f + c[1] * 0.587f + c[2] * 0.114f;
        float *o = bright + 4 * k;
        for (int ch = 0; ch < 4; ch++) o[ch] = lum > threshold ? store(c[ch], half_storage) : 0.0f;
    }
    const float *src = bright;
    uint32_t sw = hw, sh = hh;
    for (int i = 0; i < blur_passes; i++) {
        blur_pass(src, sw, sh, bw, bh, 0, half_storage, b1);
        blur_pass(b1, bw, bh, bw, bh, 1, half_storage, b2);
        src = b2;
        sw = bw;
        sh = bh;
    }
#pragma omp parallel for schedule(static)
    for (long long k = 0; k < (long long)w * h; k++) { /* combine, bloom.glsl.ts:95-127 */
        const uint32_t px = (uint32_t)(k % w), py = (uint32_t)(k / w);
        const float u = ((float)px + 0.5f) / (float)w, v = ((float)py + 0.5f) / (float)h;
        float sc[4], bl[4];
        sample_linear(scene, w, h, u, v, sc);
        sample_linear(src, sw, sh, u, v, bl);
        float *o = out + 4 * k;
        for (int c = 0; c < 3; c++) o[c] = orc_powf(aces(sc[c] + bl[c] * intensity), 0.4545f);
        o[3] = 1.0f;
    }
    free(bright);
    free(b1);
    free(b2);
}
