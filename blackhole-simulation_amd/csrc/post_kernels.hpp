// post_kernels.hpp -- the reference's post chain as HIP kernels.  Two arithmetic
// contracts, one per translation unit, as for the shader marches: STRICT (kernels_strict.hip,
// -ffp-contract=off: the shaders' operation order, IEEE divide / sqrt, specified powf) and FAST
// (kernels_fast.hip: FMA contraction, reciprocal-based divide / sqrt, v_log / v_exp gamma):
//   taa_resolve_kernel   src/shaders/postprocess/reprojection.glsl.ts:44-116
//                        (driven by src/rendering/reprojection.ts:196-262, renderScale = 1)
//   ataa_resolve_kernel  src/shaders/postprocess/ataa.wgsl.ts:29-86
//   bloom_*_kernel       src/shaders/postprocess/bloom.glsl.ts:35-127, pass sequence
//                        src/rendering/bloom.ts:443-583
// Images are RGBA f32 (float4), row-major, resident in HBM.  All five kernels are
// HBM-streaming stencils: one thread per output pixel, a wave covers a 64x1 row segment so
// every fetch of a row is one coalesced 1 KiB transaction.  The TAA resolve stages the current
// frame's 64x4 tile plus a 2-texel halo in LDS once (2.1 global fetches per pixel instead of
// 44): its eleven bilinear taps then read LDS (0.26 -> 0.19 ms at 4K).
// Texture fetches: GL LINEAR + CLAMP_TO_EDGE with f32 weights.  Render targets are RGBA16F
// upstream: with half_storage every stored channel is rounded through binary16 (RNE).
#pragma once

#include <hip/hip_fp16.h>

#include "engine_types.hpp"
#include "shader_common.hpp"

namespace {

using namespace grvhip;

__device__ __forceinline__ float post_store(float v, int half) {
    return half ? __half2float(__float2half_rn(v)) : v;
}
__device__ __forceinline__ float post_clamp(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
__device__ __forceinline__ int post_clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
__device__ __forceinline__ float post_mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }

// texture(tex, uv): LINEAR + CLAMP_TO_EDGE
__device__ __forceinline__ float4 post_sample(const float4 *__restrict__ tex, uint32_t w, uint32_t h,
                                              float u, float v) {
    const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float a = x - fx, b = y - fy;
    const int i0 = post_clampi((int)fx, 0, (int)w - 1), i1 = post_clampi((int)fx + 1, 0, (int)w - 1);
    const int j0 = post_clampi((int)fy, 0, (int)h - 1), j1 = post_clampi((int)fy + 1, 0, (int)h - 1);
    const float4 t00 = tex[(size_t)j0 * w + i0], t10 = tex[(size_t)j0 * w + i1];
    const float4 t01 = tex[(size_t)j1 * w + i0], t11 = tex[(size_t)j1 * w + i1];
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return make_float4(w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x,
                       w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y,
                       w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z,
                       w00 * t00.w + w10 * t10.w + w01 * t01.w + w11 * t11.w);
}

// texture(tex, uv) at the centre of texel (px, py) of a w x h image.  A LINEAR sampler returns the
// texel itself there; shader order still evaluates the f32 bilinear arithmetic (a weight of ~1e-7 can
// leak from a neighbour when the tap position does not round back onto the centre), the FAST
// contract reads the texel.
template <int ARITH>
__device__ __forceinline__ float4 post_fetch_centre(const float4 *__restrict__ tex, uint32_t w, uint32_t h,
                                                    uint32_t px, uint32_t py) {
    if constexpr (ARITH == GRV_ARITH_FAST) return tex[(size_t)py * w + px];
    else return post_sample(tex, w, h, ((float)px + 0.5f) / (float)w, ((float)py + 0.5f) / (float)h);
}

// ---- LDS-staged tile of the current frame: 64x4 pixels + halo -------------------------------
constexpr int kTileW = 64, kTileH = 4;
template <int HALO> struct PostTile {
    static constexpr int W = kTileW + 2 * HALO, H = kTileH + 2 * HALO;
    float4 t[H][W]; // t[0][0] is image texel (blockIdx.x * 64 - HALO, blockIdx.y * 4 - HALO)
};
// fill with edge-clamped texels (CLAMP_TO_EDGE is applied when the tile is read back by
// clamped image coordinates, so the halo only has to exist where the image does)
template <int HALO>
__device__ __forceinline__ void post_tile_load(PostTile<HALO> &T, const float4 *__restrict__ img,
                                               uint32_t w, uint32_t h) {
    const int tid = threadIdx.y * kTileW + threadIdx.x;
    const int bx0 = (int)(blockIdx.x * kTileW) - HALO, by0 = (int)(blockIdx.y * kTileH) - HALO;
    for (int k = tid; k < PostTile<HALO>::W * PostTile<HALO>::H; k += kTileW * kTileH) {
        const int ly = k / PostTile<HALO>::W, lx = k % PostTile<HALO>::W;
        const int gx = post_clampi(bx0 + lx, 0, (int)w - 1), gy = post_clampi(by0 + ly, 0, (int)h - 1);
        T.t[ly][lx] = img[(size_t)gy * w + gx];
    }
    __syncthreads();
}
// texel (i, j) of the image (already clamped to the image) through the tile.  A tap at an offset
// of at most HALO - 1 texels touches texels within HALO of the pixel (bilinear footprint), and
// clamping to the image only moves a coordinate towards the pixel: always inside the window.
template <int HALO>
__device__ __forceinline__ float4 post_tile_texel(const PostTile<HALO> &T, const float4 *__restrict__,
                                                  uint32_t, int i, int j) {
    const int lx = i - ((int)(blockIdx.x * kTileW) - HALO), ly = j - ((int)(blockIdx.y * kTileH) - HALO);
    return T.t[ly][lx];
}
// texture(tex, uv) LINEAR + CLAMP_TO_EDGE, texels through the tile: same arithmetic as post_sample
template <int HALO>
__device__ __forceinline__ float4 post_sample_tile(const PostTile<HALO> &T, const float4 *__restrict__ tex,
                                                   uint32_t w, uint32_t h, float u, float v) {
    const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float a = x - fx, b = y - fy;
    const int i0 = post_clampi((int)fx, 0, (int)w - 1), i1 = post_clampi((int)fx + 1, 0, (int)w - 1);
    const int j0 = post_clampi((int)fy, 0, (int)h - 1), j1 = post_clampi((int)fy + 1, 0, (int)h - 1);
    const float4 t00 = post_tile_texel(T, tex, w, i0, j0), t10 = post_tile_texel(T, tex, w, i1, j0);
    const float4 t01 = post_tile_texel(T, tex, w, i0, j1), t11 = post_tile_texel(T, tex, w, i1, j1);
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return make_float4(w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x,
                       w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y,
                       w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z,
                       w00 * t00.w + w10 * t10.w + w01 * t01.w + w11 * t11.w);
}

struct YCC {
    float y, co, cg;
};
__device__ __forceinline__ YCC to_ycocg(float r, float g, float b) {
    return YCC{r * 0.25f + g * 0.5f + b * 0.25f, r * 0.5f + g * 0.0f + b * -0.5f,
               r * -0.25f + g * 0.5f + b * -0.25f};
}
__device__ __forceinline__ float4 from_ycocg(YCC c, int half) {
    return make_float4(post_store(c.y + c.co - c.cg, half), post_store(c.y + c.cg, half),
                       post_store(c.y - c.co - c.cg, half), 1.0f);
}

// first and second moments of the 3x3 YCoCg neighbourhood
struct Moments {
    float m1[3], m2[3];
    __device__ __forceinline__ void add(YCC s) {
        m1[0] += s.y;
        m1[1] += s.co;
        m1[2] += s.cg;
        m2[0] += s.y * s.y;
        m2[1] += s.co * s.co;
        m2[2] += s.cg * s.cg;
    }
    __device__ __forceinline__ void mean_std(float mean[3], float sd[3]) const {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            mean[c] = m1[c] / 9.0f;
            sd[c] = sqrtf(fmaxf(m2[c] / 9.0f - mean[c] * mean[c], 0.0f));
        }
    }
};

__device__ __forceinline__ bool post_pixel(uint32_t w, uint32_t h, uint32_t &px, uint32_t &py) {
    px = blockIdx.x * blockDim.x + threadIdx.x;
    py = blockIdx.y;
    return px < w && py < h;
}

template <int ARITH>
__global__ __launch_bounds__(256) void taa_resolve_kernel(uint32_t w, uint32_t h,
                                                          const float4 *__restrict__ current,
                                                          const float4 *__restrict__ history,
                                                          float blend_factor, int camera_moving,
                                                          int half_storage, float4 *__restrict__ out) {
    __shared__ PostTile<2> tile; // taps at +-1 texel may round one texel further out
    post_tile_load(tile, current, w, h);
    const uint32_t px = blockIdx.x * kTileW + threadIdx.x, py = blockIdx.y * kTileH + threadIdx.y;
    if (px >= w || py >= h) return;
    const float tx = 1.0f / (float)w, ty = 1.0f / (float)h;
    const float u = ((float)px + 0.5f) / (float)w, v = ((float)py + 0.5f) / (float)h;
    // the ten taps on the current frame sit on texel centres (pixel centre +- one texel), where a
    // LINEAR sampler returns the texel itself.  Shader order evaluates them as the f32 bilinear
    // arithmetic the oracle defines (a weight of ~1e-7 leaks from the neighbour when (px + 0.5 +- 1)
    // / w * w - 0.5 does not round back to an integer); the FAST contract reads the texels.
    float4 cur;
    Moments M{{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
    if constexpr (ARITH == GRV_ARITH_FAST) {
        const int lx = (int)threadIdx.x + 2, ly = (int)threadIdx.y + 2; // tile origin is 2 texels out
        cur = tile.t[ly][lx];
#pragma unroll
        for (int y = -1; y <= 1; ++y)
#pragma unroll
            for (int x = -1; x <= 1; ++x) {
                const float4 s = tile.t[ly + y][lx + x]; // halo texels are edge-clamped already
                M.add(to_ycocg(s.x, s.y, s.z));
            }
    } else {
        cur = post_sample_tile(tile, current, w, h, u, v);
#pragma unroll
        for (int y = -1; y <= 1; ++y)
#pragma unroll
            for (int x = -1; x <= 1; ++x) {
                const float4 s = post_sample_tile(tile, current, w, h, u + (float)x * tx, v + (float)y * ty);
                M.add(to_ycocg(s.x, s.y, s.z));
            }
    }
    float mean[3], sd[3];
    M.mean_std(mean, sd);
    const float4 h4 = post_sample(history, w, h, u, v);
    YCC hy = to_ycocg(h4.x, h4.y, h4.z);
    hy.y = post_clamp(hy.y, mean[0] - 1.5f * sd[0], mean[0] + 1.5f * sd[0]);
    hy.co = post_clamp(hy.co, mean[1] - 1.5f * sd[1], mean[1] + 1.5f * sd[1]);
    hy.cg = post_clamp(hy.cg, mean[2] - 1.5f * sd[2], mean[2] + 1.5f * sd[2]);
    const float varianceWeight = 1.0f - post_clamp(sd[0] * 4.0f, 0.0f, 0.55f);
    const float alpha = camera_moving ? 0.0f : blend_factor * varianceWeight;
    const YCC cy = to_ycocg(cur.x, cur.y, cur.z);
    out[(size_t)py * w + px] = from_ycocg(
        YCC{post_mix(cy.y, hy.y, alpha), post_mix(cy.co, hy.co, alpha), post_mix(cy.cg, hy.cg, alpha)},
        half_storage);
}

struct AtaaCamera { // CameraUniforms fields the resolve pass reads (types.wgsl.ts:6-17)
    float inv_view[16], inv_proj[16], prev_view_proj[16], position[3];
};
__device__ __forceinline__ void post_m4v4(const float *m, float x, float y, float z, float w, float o[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = m[0 + r] * x + m[4 + r] * y + m[8 + r] * z + m[12 + r] * w;
}

template <int ARITH>
__global__ __launch_bounds__(256) void ataa_resolve_kernel(uint32_t w, uint32_t h, AtaaCamera cam,
                                                           const float4 *__restrict__ current,
                                                           const float4 *__restrict__ history,
                                                           int half_storage, float4 *__restrict__ out) {
    // nine integer texel loads per pixel: rows stay in L1/L2, an LDS stage measured slower here
    const uint32_t px = blockIdx.x * kTileW + threadIdx.x, py = blockIdx.y * kTileH + threadIdx.y;
    if (px >= w || py >= h) return;
    const float u = ((float)px + 0.5f) / (float)w, v = ((float)py + 0.5f) / (float)h;
    const float4 c4 = current[(size_t)py * w + px];
    const YCC center = to_ycocg(c4.x, c4.y, c4.z);
    Moments M{{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int sx = post_clampi((int)px + dx, 0, (int)w - 1), sy = post_clampi((int)py + dy, 0, (int)h - 1);
            const float4 s = current[(size_t)sy * w + sx];
            M.add(to_ycocg(s.x, s.y, s.z));
        }
    float mean[3], sd[3];
    M.mean_std(mean, sd);
    const float ndcx = u * 2.0f - 1.0f, ndcy = v * 2.0f - 1.0f;
    float vt[4], wd[4], pc[4];
    post_m4v4(cam.inv_proj, ndcx, -ndcy, 1.0f, 1.0f, vt);
    float vx = vt[0] / vt[3], vy = vt[1] / vt[3], vz = vt[2] / vt[3];
    const float len = sqrtf(vx * vx + vy * vy + vz * vz);
    vx /= len;
    vy /= len;
    vz /= len;
    post_m4v4(cam.inv_view, vx, vy, vz, 0.0f, wd);
    const float depth = 12.0f; // reprojectDepth, ataa.wgsl.ts:68
    post_m4v4(cam.prev_view_proj, cam.position[0] + wd[0] * depth, cam.position[1] + wd[1] * depth,
              cam.position[2] + wd[2] * depth, 1.0f, pc);
    const float pu = (pc[0] / pc[3]) * 0.5f + 0.5f, pv = (pc[1] / pc[3]) * -0.5f + 0.5f;
    const float4 h4 = post_sample(history, w, h, pu, pv);
    YCC hy = to_ycocg(h4.x, h4.y, h4.z);
    hy.y = post_clamp(hy.y, mean[0] - 2.0f * sd[0], mean[0] + 2.0f * sd[0]);
    hy.co = post_clamp(hy.co, mean[1] - 2.0f * sd[1], mean[1] + 2.0f * sd[1]);
    hy.cg = post_clamp(hy.cg, mean[2] - 2.0f * sd[2], mean[2] + 2.0f * sd[2]);
    out[(size_t)py * w + px] = from_ycocg(YCC{post_mix(center.y, hy.y, 0.92f), post_mix(center.co, hy.co, 0.92f),
                                              post_mix(center.cg, hy.cg, 0.92f)},
                                          half_storage);
}

// bloom.glsl.ts:35-58: dst (dw x dh) <- bright pixels of src (w x h)
template <int ARITH>
__global__ __launch_bounds__(256) void bloom_bright_kernel(uint32_t w, uint32_t h,
                                                           const float4 *__restrict__ src, uint32_t dw,
                                                           uint32_t dh, float threshold, int half_storage,
                                                           float4 *__restrict__ dst) {
    uint32_t px, py;
    if (!post_pixel(dw, dh, px, py)) return;
    const float4 c = post_sample(src, w, h, ((float)px + 0.5f) / (float)dw, ((float)py + 0.5f) / (float)dh);
    const float lum = c.x * 0.299f + c.y * 0.587f + c.z * 0.114f;
    dst[(size_t)py * dw + px] =
        lum > threshold ? make_float4(post_store(c.x, half_storage), post_store(c.y, half_storage),
                                      post_store(c.z, half_storage), post_store(c.w, half_storage))
                        : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// bloom.glsl.ts:64-89: 9-tap separable Gaussian; u_resolution = destination size
template <int ARITH>
__global__ __launch_bounds__(256) void bloom_blur_kernel(uint32_t sw, uint32_t sh,
                                                         const float4 *__restrict__ src, uint32_t dw,
                                                         uint32_t dh, int vertical, int half_storage,
                                                         float4 *__restrict__ dst) {
    uint32_t px, py;
    if (!post_pixel(dw, dh, px, py)) return;
    const float wts[5] = {0.227027f, 0.1945946f, 0.1216216f, 0.054054f, 0.016216f};
    if constexpr (ARITH == GRV_ARITH_FAST) {
        if (sw == dw && sh == dh) { // every tap sits on a texel centre: nine texel reads
            const float4 c = src[(size_t)py * sw + px];
            float r = c.x * wts[0], g = c.y * wts[0], b = c.z * wts[0];
#pragma unroll
            for (int i = 1; i < 5; ++i) {
                const int xa = vertical ? (int)px : post_clampi((int)px + i, 0, (int)sw - 1);
                const int xb = vertical ? (int)px : post_clampi((int)px - i, 0, (int)sw - 1);
                const int ya = vertical ? post_clampi((int)py + i, 0, (int)sh - 1) : (int)py;
                const int yb = vertical ? post_clampi((int)py - i, 0, (int)sh - 1) : (int)py;
                const float4 p = src[(size_t)ya * sw + xa], m = src[(size_t)yb * sw + xb];
                r += (p.x + m.x) * wts[i];
                g += (p.y + m.y) * wts[i];
                b += (p.z + m.z) * wts[i];
            }
            dst[(size_t)py * dw + px] = make_float4(post_store(r, half_storage), post_store(g, half_storage),
                                                    post_store(b, half_storage), 1.0f);
            return;
        }
    }
    const float tx = 1.0f / (float)dw, ty = 1.0f / (float)dh;
    const float u = ((float)px + 0.5f) / (float)dw, v = ((float)py + 0.5f) / (float)dh;
    const float4 c0 = post_sample(src, sw, sh, u, v);
    float r = c0.x * wts[0], g = c0.y * wts[0], b = c0.z * wts[0];
#pragma unroll
    for (int i = 1; i < 5; ++i) {
        const float ox = (vertical ? 0.0f : 1.0f) * tx * (float)i, oy = (vertical ? 1.0f : 0.0f) * ty * (float)i;
        const float4 p = post_sample(src, sw, sh, u + ox, v + oy);
        const float4 m = post_sample(src, sw, sh, u - ox, v - oy);
        r += p.x * wts[i];
        r += m.x * wts[i];
        g += p.y * wts[i];
        g += m.y * wts[i];
        b += p.z * wts[i];
        b += m.z * wts[i];
    }
    dst[(size_t)py * dw + px] =
        make_float4(post_store(r, half_storage), post_store(g, half_storage), post_store(b, half_storage), 1.0f);
}

// gamma: the specified powf (sh_powf) in shader order; v_log_f32 / v_exp_f32 in the FAST contract
template <int ARITH> __device__ __forceinline__ float post_pow(float x, float y) {
    if constexpr (ARITH == GRV_ARITH_FAST) return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
    else return sh_powf(x, y);
}
__device__ __forceinline__ float post_aces(float c) {
    return post_clamp((c * (2.51f * c + 0.03f)) / (c * (2.43f * c + 0.59f) + 0.14f), 0.0f, 1.0f);
}

// bloom.glsl.ts:95-127: scene + bloom * intensity -> ACES -> gamma (the backbuffer)
template <int ARITH>
__global__ __launch_bounds__(256) void bloom_combine_kernel(uint32_t w, uint32_t h,
                                                            const float4 *__restrict__ scene, uint32_t bw,
                                                            uint32_t bh, const float4 *__restrict__ bloom,
                                                            float intensity, float4 *__restrict__ out) {
    uint32_t px, py;
    if (!post_pixel(w, h, px, py)) return;
    const float u = ((float)px + 0.5f) / (float)w, v = ((float)py + 0.5f) / (float)h;
    const float4 s = post_fetch_centre<ARITH>(scene, w, h, px, py);
    const float4 b = post_sample(bloom, bw, bh, u, v);
    out[(size_t)py * w + px] = make_float4(post_pow<ARITH>(post_aces(s.x + b.x * intensity), 0.4545f),
                                           post_pow<ARITH>(post_aces(s.y + b.y * intensity), 0.4545f),
                                           post_pow<ARITH>(post_aces(s.z + b.z * intensity), 0.4545f), 1.0f);
}

// ---- fused bloom passes ------------------------------------------------------------------------
// BloomManager.applyBloomToTexture (bloom.ts:443-583) is bright -> passes x (H, V) -> combine, every
// pass a full round trip of its render target.  Two of those round trips carry no reuse across the
// image and are fused away when the sizes nest (w = 4 bw, h = 4 bh, hw = 2 bw, hh = 2 bh):
//   bright + first H blur : the block builds the thresholded half-resolution tile it needs in LDS
//                           straight from the scene (the same per-texel expression as
//                           bloom_bright_kernel) and blurs out of LDS -- the half-resolution bright
//                           target is never written or read;
//   last V blur + combine : the block blurs the 18 x 6 quarter-resolution texels its 64 x 16 output
//                           pixels will tap into LDS, then combines -- the last blur target is never
//                           written or read.
// Both evaluate the very expressions of the separate kernels (post_sample's arithmetic through a
// window accessor), so the STRICT forms stay bit-identical to the unfused chain and to the oracle.
struct LdsWindow {      // texel (i, j) of an image, i in [x0, x0 + W), j in [y0, y0 + H), row pitch W
    const float4 *t;
    int x0, y0, W;
    __device__ __forceinline__ float4 at(int i, int j) const { return t[(j - y0) * W + (i - x0)]; }
};
// texture(tex, uv) LINEAR + CLAMP_TO_EDGE on a w x h image whose texels come from the window:
// post_sample's arithmetic, texel for texel
__device__ __forceinline__ float4 post_sample_win(const LdsWindow &L, uint32_t w, uint32_t h, float u, float v) {
    const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float a = x - fx, b = y - fy;
    const int i0 = post_clampi((int)fx, 0, (int)w - 1), i1 = post_clampi((int)fx + 1, 0, (int)w - 1);
    const int j0 = post_clampi((int)fy, 0, (int)h - 1), j1 = post_clampi((int)fy + 1, 0, (int)h - 1);
    const float4 t00 = L.at(i0, j0), t10 = L.at(i1, j0), t01 = L.at(i0, j1), t11 = L.at(i1, j1);
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return make_float4(w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x,
                       w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y,
                       w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z,
                       w00 * t00.w + w10 * t10.w + w01 * t01.w + w11 * t11.w);
}
// the 9-tap blur of bloom.glsl.ts:64-89 at (u, v) over a windowed source
__device__ __forceinline__ float4 post_blur9_win(const LdsWindow &L, uint32_t sw, uint32_t sh, float u, float v,
                                                 float tx, float ty, int vertical, int half_storage) {
    const float wts[5] = {0.227027f, 0.1945946f, 0.1216216f, 0.054054f, 0.016216f};
    const float4 c0 = post_sample_win(L, sw, sh, u, v);
    float r = c0.x * wts[0], g = c0.y * wts[0], b = c0.z * wts[0];
#pragma unroll
    for (int i = 1; i < 5; ++i) {
        const float ox = (vertical ? 0.0f : 1.0f) * tx * (float)i, oy = (vertical ? 1.0f : 0.0f) * ty * (float)i;
        const float4 p = post_sample_win(L, sw, sh, u + ox, v + oy);
        const float4 m = post_sample_win(L, sw, sh, u - ox, v - oy);
        r += p.x * wts[i];
        r += m.x * wts[i];
        g += p.y * wts[i];
        g += m.y * wts[i];
        b += p.z * wts[i];
        b += m.z * wts[i];
    }
    return make_float4(post_store(r, half_storage), post_store(g, half_storage), post_store(b, half_storage), 1.0f);
}

// bright pass + first horizontal blur.  Block = 64 x 4 texels of the quarter-size target; with
// hw = 2 bw a tap at (px + k) reads bright texels 2 (px + k) and 2 (px + k) + 1 (one more on either
// side is kept for f32 rounding of the tap position), rows 2 py and 2 py + 1.
constexpr int kBrightWinW = 2 * (kTileW + 8) + 4, kBrightWinH = 2 * kTileH + 2;
template <int ARITH>
__global__ __launch_bounds__(256) void bloom_bright_hblur_kernel(uint32_t w, uint32_t h,
                                                                 const float4 *__restrict__ scene, uint32_t hw,
                                                                 uint32_t hh, uint32_t bw, uint32_t bh,
                                                                 float threshold, int half_storage,
                                                                 float4 *__restrict__ dst) {
    __shared__ float4 win[kBrightWinH * kBrightWinW];
    const int x0 = 2 * ((int)(blockIdx.x * kTileW) - 4) - 2, y0 = 2 * (int)(blockIdx.y * kTileH) - 1;
    const int tid = threadIdx.y * kTileW + threadIdx.x;
    for (int k = tid; k < kBrightWinW * kBrightWinH; k += kTileW * kTileH) {
        const int gx = x0 + k % kBrightWinW, gy = y0 + k / kBrightWinW;
        float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (gx >= 0 && gy >= 0 && gx < (int)hw && gy < (int)hh) { // bloom_bright_kernel's texel (gx, gy)
            const float4 c = post_sample(scene, w, h, ((float)gx + 0.5f) / (float)hw, ((float)gy + 0.5f) / (float)hh);
            const float lum = c.x * 0.299f + c.y * 0.587f + c.z * 0.114f;
            if (lum > threshold)
                o = make_float4(post_store(c.x, half_storage), post_store(c.y, half_storage),
                                post_store(c.z, half_storage), post_store(c.w, half_storage));
        }
        win[k] = o;
    }
    __syncthreads();
    const uint32_t px = blockIdx.x * kTileW + threadIdx.x, py = blockIdx.y * kTileH + threadIdx.y;
    if (px >= bw || py >= bh) return;
    const LdsWindow L{win, x0, y0, kBrightWinW};
    dst[(size_t)py * bw + px] = post_blur9_win(L, hw, hh, ((float)px + 0.5f) / (float)bw, ((float)py + 0.5f) / (float)bh,
                                               1.0f / (float)bw, 1.0f / (float)bh, 0, half_storage);
}

// a middle (V, H) pair of the blur chain in one launch: the block blurs the 74 x 6 texels its
// 64 x 4 outputs tap (4 texels either side along x, one more for f32 rounding of the tap position)
// vertically into LDS, then horizontally out of it -- the intermediate target of the pair stays on
// chip.  Both steps are bloom_blur_kernel's expressions.
constexpr int kMvW = kTileW + 10, kMvH = kTileH + 2, kMsW = kMvW + 2, kMsH = kMvH + 10;
template <int ARITH>
__global__ __launch_bounds__(256) void bloom_vh_blur_kernel(uint32_t bw, uint32_t bh,
                                                            const float4 *__restrict__ src, int half_storage,
                                                            float4 *__restrict__ dst) {
    __shared__ float4 src_s[kMsH * kMsW];
    __shared__ float4 v_s[kMvH * kMvW];
    const int vx0 = (int)(blockIdx.x * kTileW) - 5, vy0 = (int)(blockIdx.y * kTileH) - 1;
    const int sx0 = vx0 - 1, sy0 = vy0 - 5;
    const int tid = threadIdx.y * kTileW + threadIdx.x;
    for (int k = tid; k < kMsW * kMsH; k += kTileW * kTileH) {
        const int gx = sx0 + k % kMsW, gy = sy0 + k / kMsW;
        src_s[k] = (gx >= 0 && gy >= 0 && gx < (int)bw && gy < (int)bh) ? src[(size_t)gy * bw + gx]
                                                                        : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    __syncthreads();
    const float tx = 1.0f / (float)bw, ty = 1.0f / (float)bh;
    const LdsWindow S{src_s, sx0, sy0, kMsW};
    for (int k = tid; k < kMvW * kMvH; k += kTileW * kTileH) {
        const int gx = vx0 + k % kMvW, gy = vy0 + k / kMvW;
        float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (gx >= 0 && gy >= 0 && gx < (int)bw && gy < (int)bh)
            o = post_blur9_win(S, bw, bh, ((float)gx + 0.5f) / (float)bw, ((float)gy + 0.5f) / (float)bh, tx, ty, 1,
                               half_storage);
        v_s[k] = o;
    }
    __syncthreads();
    const uint32_t px = blockIdx.x * kTileW + threadIdx.x, py = blockIdx.y * kTileH + threadIdx.y;
    if (px >= bw || py >= bh) return;
    const LdsWindow V{v_s, vx0, vy0, kMvW};
    dst[(size_t)py * bw + px] = post_blur9_win(V, bw, bh, ((float)px + 0.5f) / (float)bw, ((float)py + 0.5f) / (float)bh,
                                               tx, ty, 0, half_storage);
}

// last vertical blur + combine.  Block = 64 x 16 output pixels = 16 x 4 quarter-size texels; the
// combine's bilinear taps reach one texel further on every side (18 x 6), each of those is a 9-tap
// vertical blur whose own taps may round one texel sideways (source window 20 x 16).
constexpr int kVbW = kTileW / 4 + 2, kVbH = 4 + 2, kVsW = kVbW + 2, kVsH = kVbH + 10;
template <int ARITH>
__global__ __launch_bounds__(256) void bloom_vblur_combine_kernel(uint32_t w, uint32_t h,
                                                                  const float4 *__restrict__ scene, uint32_t bw,
                                                                  uint32_t bh, const float4 *__restrict__ hblur,
                                                                  float intensity, int half_storage,
                                                                  float4 *__restrict__ out) {
    __shared__ float4 src_s[kVsH * kVsW];
    __shared__ float4 vb_s[kVbH * kVbW];
    const int qx0 = (int)(blockIdx.x * (kTileW / 4)), qy0 = (int)(blockIdx.y * 4);
    const int sx0 = qx0 - 2, sy0 = qy0 - 6, vx0 = qx0 - 1, vy0 = qy0 - 1;
    const int tid = threadIdx.y * kTileW + threadIdx.x;
    for (int k = tid; k < kVsW * kVsH; k += kTileW * kTileH) {
        const int gx = sx0 + k % kVsW, gy = sy0 + k / kVsW;
        src_s[k] = (gx >= 0 && gy >= 0 && gx < (int)bw && gy < (int)bh) ? hblur[(size_t)gy * bw + gx]
                                                                        : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    __syncthreads();
    if (tid < kVbW * kVbH) {
        const int gx = vx0 + tid % kVbW, gy = vy0 + tid / kVbW;
        float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (gx >= 0 && gy >= 0 && gx < (int)bw && gy < (int)bh) { // bloom_blur_kernel's (vertical) texel (gx, gy)
            const LdsWindow L{src_s, sx0, sy0, kVsW};
            o = post_blur9_win(L, bw, bh, ((float)gx + 0.5f) / (float)bw, ((float)gy + 0.5f) / (float)bh,
                               1.0f / (float)bw, 1.0f / (float)bh, 1, half_storage);
        }
        vb_s[tid] = o;
    }
    __syncthreads();
    const LdsWindow V{vb_s, vx0, vy0, kVbW};
    const uint32_t px = blockIdx.x * kTileW + threadIdx.x;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t py = blockIdx.y * 16u + (uint32_t)r * 4u + threadIdx.y;
        if (px >= w || py >= h) continue;
        const float u = ((float)px + 0.5f) / (float)w, v = ((float)py + 0.5f) / (float)h;
        const float4 s = post_fetch_centre<ARITH>(scene, w, h, px, py);
        const float4 b = post_sample_win(V, bw, bh, u, v);
        out[(size_t)py * w + px] = make_float4(post_pow<ARITH>(post_aces(s.x + b.x * intensity), 0.4545f),
                                               post_pow<ARITH>(post_aces(s.y + b.y * intensity), 0.4545f),
                                               post_pow<ARITH>(post_aces(s.z + b.z * intensity), 0.4545f), 1.0f);
    }
}

// A compute / fragment pass that writes an RGBA16F target: round the stored channels in place
// (the march kernels keep f32 outputs; the renderer layer applies the storage format).
__global__ __launch_bounds__(256) void post_quantize_kernel(float4 *__restrict__ img, uint32_t n_px) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_px) return;
    const float4 c = img[k];
    img[k] = make_float4(post_store(c.x, 1), post_store(c.y, 1), post_store(c.z, 1), post_store(c.w, 1));
}

// blit pass of the WebGPU renderer (src/rendering/webgpu/renderer.ts:14-50): the resolved
// history sampled at the pixel centre, Reinhard c / (c + 1), alpha passed through
template <int ARITH>
__global__ __launch_bounds__(256) void blit_reinhard_kernel(uint32_t w, uint32_t h,
                                                            const float4 *__restrict__ src,
                                                            float4 *__restrict__ dst) {
    uint32_t px, py;
    if (!post_pixel(w, h, px, py)) return;
    // the blit samples the resolve target 1:1 at pixel centres, where a linear sampler returns
    // the texel itself (8-bit fixed-point weights): a plain fetch
    const float4 c = src[(size_t)py * w + px];
    dst[(size_t)py * w + px] = make_float4(c.x / (c.x + 1.0f), c.y / (c.y + 1.0f), c.z / (c.z + 1.0f), c.w);
}

} // namespace
