// engine_shaders.hip -- f32 shader frames (WGSL compute march, GLSL fragment shader), the post
// chain and the two renderers' render() sequences behind the C ABI.  See engine_internal.hpp.
#include "engine_internal.hpp"
#include "strict_libm.hpp"

using namespace grvhost;

extern "C" {

void grv_wgsl_params_default(uint32_t width, uint32_t height, const GrvCamera *cam, double mass,
                             double spin, GrvWgslParams *p) {
    if (!p) return;
    std::memset(p, 0, sizeof *p);
    p->width = width;
    p->height = height;
    if (cam) {
        for (int k = 0; k < 16; ++k) {
            p->inv_view[k] = (float)cam->inv_view[k];
            p->inv_proj[k] = (float)cam->inv_proj[k];
        }
        for (int k = 0; k < 3; ++k) p->position[k] = (float)cam->position[k];
    }
    p->mass = (float)mass;
    p->spin = (float)spin;
    p->max_steps = 150; // compute.wgsl.ts:13
    p->tile_world = 1;
    p->stars = 1;
}

void grv_glsl_params_default(uint32_t width, uint32_t height, double mass, double spin,
                             GrvGlslParams *p) {
    if (!p) return;
    std::memset(p, 0, sizeof *p);
    p->width = width;
    p->height = height;
    p->mass = (float)mass;
    p->spin = (float)(spin * mass);              // renderer.ts:326
    p->zoom = 30.0f * 2.0f;                       // simulation.config.ts:118-119, renderer.ts:327
    p->mouse[0] = 0.5f;
    p->mouse[1] = 97.0f / 180.0f;                 // simulation.config.ts:106-107
    p->disk_size = 50.0f;                         // simulation.config.ts:138-139
    p->disk_scale_height = 0.2f;                  // :147-148
    p->disk_density = 4.0f;                       // :167-168
    p->disk_temp = (float)(9500.0 * strictm::sl_pow(mass, -0.25)); // renderer.ts:352-356
    p->lensing_strength = 1.0f;                   // renderer.ts:340
    p->time = 0.0f;
    p->turbulence = -1.0f;                        // sample the noise texture (disk.ts:55)
    p->max_ray_steps = 256;                       // simulation.config.ts:205-211 (ultra)
    p->tone_map = 0;
    p->tile_world = 1;
    p->features = GRV_GLSL_FEATURES_DEFAULT;
    p->quality = 1;
    p->cam_quat[3] = 1.0f;                        // renderer.ts:315-316
}

void grv_seeded_noise_rgba8(uint32_t seed, uint32_t size, uint8_t *rgba) {
    if (!rgba) return;
    // xorshift32 stream; byte = floor(u * 255), u in [0, 1), as createNoiseTexture forms it
    uint32_t x = seed ? seed : 0x9E3779B9u;
    const size_t n = (size_t)size * size * 4u;
    for (size_t i = 0; i < n; ++i) {
        x ^= x << 13;
        x ^= x >> 17;
        x ^= x << 5;
        rgba[i] = (uint8_t)std::floor((double)(x >> 8) / 16777216.0 * 255.0);
    }
}

int grv_set_glsl_noise(grv_engine *e, const uint8_t *noise_rgba, const uint8_t *blue_rgba) {
    if (!e) return GRV_ERR_INVALID;
    GRV_HIP(e, hipSetDevice(e->device));
    constexpr size_t kPlane = 256 * 256;
    // [2][kPlane] bytes (noise, blue noise), then the noise plane once more as f32 texel values
    if (!e->d_noise) {
        GRV_HIP(e, hipMalloc(reinterpret_cast<void **>(&e->d_noise), 2 * kPlane + kPlane * sizeof(float)));
        GRV_HIP(e, hipMemset(e->d_noise, 0, 2 * kPlane + kPlane * sizeof(float)));
    }
    std::vector<uint8_t> plane(kPlane);
    const uint8_t *src[2] = {noise_rgba, blue_rgba};
    for (int t = 0; t < 2; ++t) {
        if (!src[t]) continue;
        for (size_t i = 0; i < kPlane; ++i) plane[i] = src[t][4 * i]; // .r
        GRV_HIP(e, hipMemcpy(e->d_noise + t * kPlane, plane.data(), kPlane, hipMemcpyHostToDevice));
        if (t == 0) { // UNORM8 -> float as the sampler does it: byte / 255.0f, the IEEE quotient
            std::vector<float> tex(kPlane);
            for (size_t i = 0; i < kPlane; ++i) tex[i] = (float)plane[i] / 255.0f;
            GRV_HIP(e, hipMemcpy(e->d_noise + 2 * kPlane, tex.data(), kPlane * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    return GRV_OK;
}


int grv_render_frame_wgsl(grv_engine *e, const GrvWgslParams *p, float *d_rgba, uint32_t *d_steps,
                          uint64_t *total_steps, void *stream) {
    if (!e) return GRV_ERR_INVALID;
    if (!p || !d_rgba) return fail(e, GRV_ERR_INVALID, "null argument");
    if (p->arith < GRV_ARITH_STRICT || p->arith > GRV_ARITH_FAST_PACKED)
        return fail(e, GRV_ERR_INVALID, "invalid arith %d", p->arith);
    WgslParams P{};
    std::memcpy(P.inv_view, p->inv_view, sizeof P.inv_view);
    std::memcpy(P.inv_proj, p->inv_proj, sizeof P.inv_proj);
    std::memcpy(P.position, p->position, sizeof P.position);
    P.mass = p->mass;
    P.spin = p->spin;
    P.jitter[0] = p->jitter[0];
    P.jitter[1] = p->jitter[1];
    P.max_steps = p->max_steps;
    P.stars = p->stars;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return run_shader_frame(e, p->width, p->height, p->tile_world, p->tile_rank, total_steps, s, 1,
                            p->arith == GRV_ARITH_FAST_PACKED ? march_blocks_pk : nullptr, P.max_steps,
                            [&](const FrameGeom &G, uint32_t n, unsigned long long *tot, MarchSched sched) {
                                if (p->arith == GRV_ARITH_FAST_PACKED)
                                    return launch_wgsl_symplectic_pk(G, P, d_rgba, d_steps, tot, n, sched, s);
                                return p->arith == GRV_ARITH_FAST
                                           ? launch_wgsl_symplectic_fast(G, P, d_rgba, d_steps, tot, n, s)
                                           : launch_wgsl_symplectic(G, P, d_rgba, d_steps, tot, n, s);
                            });
}

int grv_render_frame_glsl(grv_engine *e, const GrvGlslParams *p, float *d_rgba, uint32_t *d_steps,
                          uint64_t *total_steps, void *stream) {
    if (!e) return GRV_ERR_INVALID;
    if (!p || !d_rgba) return fail(e, GRV_ERR_INVALID, "null argument");
    if (p->arith != GRV_ARITH_STRICT && p->arith != GRV_ARITH_FAST)
        return fail(e, GRV_ERR_INVALID, "invalid arith %d", p->arith);
    GlslParams P{};
    P.mass = p->mass;
    P.spin = p->spin;
    P.zoom = p->zoom;
    P.mouse[0] = p->mouse[0];
    P.mouse[1] = p->mouse[1];
    P.disk_size = p->disk_size;
    P.disk_scale_height = p->disk_scale_height;
    P.disk_density = p->disk_density;
    P.disk_temp = p->disk_temp;
    P.lensing_strength = p->lensing_strength;
    P.time = p->time;
    P.turbulence = p->turbulence;
    P.max_ray_steps = p->max_ray_steps;
    P.tone_map = p->tone_map;
    P.features = p->features;
    P.quality = p->quality;
    P.show_redshift = p->show_redshift;
    P.show_kerr_shadow = p->show_kerr_shadow;
    P.debug = p->debug;
    std::memcpy(P.cam_pos, p->cam_pos, sizeof P.cam_pos);
    std::memcpy(P.cam_quat, p->cam_quat, sizeof P.cam_quat);
    P.shadow_count = p->shadow_count;
    std::memcpy(P.shadow_curve, p->shadow_curve, sizeof P.shadow_curve);
    if (!e->d_noise) { // first GLSL frame of this engine: the seeded default textures
        std::vector<uint8_t> a(256 * 256 * 4), b(256 * 256 * 4);
        grv_seeded_noise_rgba8(1u, 256, a.data());
        grv_seeded_noise_rgba8(2u, 256, b.data());
        int rc = grv_set_glsl_noise(e, a.data(), b.data());
        if (rc != GRV_OK) return rc;
    }
    P.noise_r = e->d_noise;
    P.blue_r = e->d_noise + 256 * 256;
    P.noise_f = reinterpret_cast<const float *>(e->d_noise + 2 * 256 * 256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    return run_shader_frame(e, p->width, p->height, p->tile_world, p->tile_rank, total_steps, s, 0,
                            p->arith == GRV_ARITH_FAST ? +[](uint32_t n, int32_t) { return march_blocks_glsl(n); } : nullptr, 0,
                            [&](const FrameGeom &G, uint32_t n, unsigned long long *tot, MarchSched sched) {
                                return p->arith == GRV_ARITH_FAST
                                           ? launch_glsl_fragment_fast(G, P, d_rgba, d_steps, tot, n, sched, s)
                                           : launch_glsl_fragment(G, P, d_rgba, d_steps, tot, n, s);
                            });
}

float grv_taa_effective_blend(float blend_factor, float v) {
    if (v > 0.001f) return std::fmax(0.05f, std::fmin(0.9f, 0.9f - v * 6.0f));
    return blend_factor;
}

int grv_post_taa_resolve(grv_engine *e, const GrvTaaParams *p, const float *d_current,
                         const float *d_history, float *d_out, void *stream) {
    if (!e) return GRV_ERR_INVALID;
    if (!p || !d_current || !d_history || !d_out) return fail(e, GRV_ERR_INVALID, "null argument");
    if (d_out == d_current || d_out == d_history) return fail(e, GRV_ERR_INVALID, "taa: out aliases an input");
    GRV_HIP(e, hipSetDevice(e->device));
    GRV_HIP(e, (p->arith == GRV_ARITH_FAST ? launch_taa_resolve_fast : launch_taa_resolve)(
                   p->width, p->height, d_current, d_history, p->blend_factor, p->camera_moving,
                   p->half_storage, d_out, static_cast<hipStream_t>(stream)));
    return GRV_OK;
}

int grv_post_ataa_resolve(grv_engine *e, const GrvAtaaParams *p, const float *d_current,
                          const float *d_history, float *d_out, void *stream) {
    if (!e) return GRV_ERR_INVALID;
    if (!p || !d_current || !d_history || !d_out) return fail(e, GRV_ERR_INVALID, "null argument");
    if (d_out == d_current || d_out == d_history) return fail(e, GRV_ERR_INVALID, "ataa: out aliases an input");
    AtaaCameraHost cam;
    std::memcpy(cam.inv_view, p->inv_view, sizeof cam.inv_view);
    std::memcpy(cam.inv_proj, p->inv_proj, sizeof cam.inv_proj);
    std::memcpy(cam.prev_view_proj, p->prev_view_proj, sizeof cam.prev_view_proj);
    std::memcpy(cam.position, p->position, sizeof cam.position);
    GRV_HIP(e, hipSetDevice(e->device));
    GRV_HIP(e, (p->arith == GRV_ARITH_FAST ? launch_ataa_resolve_fast : launch_ataa_resolve)(
                   p->width, p->height, cam, d_current, d_history, p->half_storage, d_out,
                   static_cast<hipStream_t>(stream)));
    return GRV_OK;
}

int grv_ataa_reproj_fold(const GrvAtaaParams *p, float out24[24]) {
    if (!p || !out24 || p->width == 0 || p->height == 0) return GRV_ERR_INVALID;
    AtaaCameraHost cam;
    std::memcpy(cam.inv_view, p->inv_view, sizeof cam.inv_view);
    std::memcpy(cam.inv_proj, p->inv_proj, sizeof cam.inv_proj);
    std::memcpy(cam.prev_view_proj, p->prev_view_proj, sizeof cam.prev_view_proj);
    std::memcpy(cam.position, p->position, sizeof cam.position);
    AtaaReprojHost rp;
    ataa_reproj_fold(cam, p->width, p->height, rp);
    std::memcpy(out24, &rp, sizeof rp);
    static_assert(sizeof(AtaaReprojHost) == 24 * sizeof(float), "24 coefficients");
    return GRV_OK;
}

void grv_bloom_params_default(uint32_t width, uint32_t height, GrvBloomParams *p) {
    if (!p) return;
    p->width = width;
    p->height = height;
    p->intensity = 0.5f; // bloom.ts:34-39
    p->threshold = 0.8f;
    p->blur_passes = 2;
    p->half_storage = 1;
    p->arith = GRV_ARITH_STRICT;
}

int grv_post_bloom(grv_engine *e, const GrvBloomParams *p, const float *d_scene, float *d_out,
                   void *stream) {
    if (!e) return GRV_ERR_INVALID;
    if (!p || !d_scene || !d_out) return fail(e, GRV_ERR_INVALID, "null argument");
    if (p->blur_passes < 0 || p->blur_passes > 64) return fail(e, GRV_ERR_INVALID, "blur_passes out of range");
    if (d_out == d_scene) return fail(e, GRV_ERR_INVALID, "bloom: out aliases the scene");
    GRV_HIP(e, hipSetDevice(e->device));
    const size_t need = bloom_scratch_floats(p->width, p->height) * sizeof(float);
    if (need > e->post_bytes) {
        if (e->post_mem) (void)hipFree(e->post_mem);
        e->post_mem = nullptr;
        e->post_bytes = 0;
        GRV_HIP(e, hipMalloc(&e->post_mem, need));
        e->post_bytes = need;
    }
    GRV_HIP(e, (p->arith == GRV_ARITH_FAST ? launch_bloom_fast : launch_bloom)(
                   p->width, p->height, d_scene, p->threshold, p->intensity, p->blur_passes, p->half_storage,
                   static_cast<float *>(e->post_mem), d_out, static_cast<hipStream_t>(stream)));
    return GRV_OK;
}

// ---- renderer layer ----
namespace {
int ensure_targets(grv_engine *e, uint32_t w, uint32_t h, hipStream_t s) {
    if (e->rt.mem && e->rt.w == w && e->rt.h == h) return GRV_OK;
    // resize: the reference recreates (zeroed) textures but keeps its frame counter and history
    // index (webgpu/renderer.ts:269-278 initTextures, reprojection.ts:102-117), so the Halton
    // jitter sequence runs on across a resolution change
    const uint32_t frames = e->rt.frames, hist = e->rt.hist;
    if (e->rt.mem) (void)hipFree(e->rt.mem);
    e->rt = grv_engine::Targets{};
    e->rt.frames = frames;
    e->rt.hist = hist;
    const size_t bytes = (size_t)3 * w * h * 4 * sizeof(float);
    GRV_HIP(e, hipMalloc(reinterpret_cast<void **>(&e->rt.mem), bytes));
    GRV_HIP(e, hipMemsetAsync(e->rt.mem, 0, bytes, s)); // textures start zeroed
    e->rt.w = w;
    e->rt.h = h;
    return GRV_OK;
}
int ensure_bloom_scratch(grv_engine *e, uint32_t w, uint32_t h, hipStream_t s) {
    const size_t need = bloom_scratch_floats(w, h) * sizeof(float);
    if (need <= e->post_bytes) return GRV_OK;
    if (e->post_mem) (void)hipFree(e->post_mem);
    e->post_mem = nullptr;
    e->post_bytes = 0;
    GRV_HIP(e, hipMalloc(&e->post_mem, need));
    GRV_HIP(e, hipMemsetAsync(e->post_mem, 0, need, s));
    e->post_bytes = need;
    return GRV_OK;
}
// halton(index, base), compute.wgsl.ts:134-145, in f32
float halton_f32(uint32_t index, uint32_t base) {
    float result = 0.0f, f = 1.0f / (float)base;
    for (uint32_t i = index; i > 0u; i /= base) {
        result += f * (float)(i % base);
        f = f / (float)base;
    }
    return result;
}
} // namespace

void grv_renderer_reset(grv_engine *e) {
    if (!e) return;
    if (e->rt.mem) (void)hipFree(e->rt.mem);
    e->rt = grv_engine::Targets{};
}
uint32_t grv_renderer_frame_count(const grv_engine *e) { return e ? e->rt.frames : 0u; }

int grv_webgpu_render(grv_engine *e, const float *cu, const float *pp, int32_t max_steps, int32_t arith,
                      float *d_screen, void *stream) {
    if (!e) return GRV_ERR_INVALID;
    if (!cu || !pp || !d_screen) return fail(e, GRV_ERR_INVALID, "null argument");
    const uint32_t w = (uint32_t)pp[2], h = (uint32_t)pp[3]; // u32(physics.resolution), compute.wgsl.ts:149-150
    if (w == 0 || h == 0 || (uint64_t)w * h > (1ull << 27)) return fail(e, GRV_ERR_INVALID, "bad resolution");
    hipStream_t s = static_cast<hipStream_t>(stream);
    GRV_HIP(e, hipSetDevice(e->device));
    int rc = ensure_targets(e, w, h, s);
    if (rc != GRV_OK) return rc;
    const size_t plane = (size_t)w * h * 4;
    float *compute_tex = e->rt.mem, *hist[2] = {e->rt.mem + plane, e->rt.mem + 2 * plane};
    // Pass 1: main ray march.  CameraUniforms floats: inv_view 32..47, inv_proj 48..63,
    // prev_view_proj 64..79, position 80..82 (types/webgpu.ts:95-116)
    GrvWgslParams wp;
    std::memset(&wp, 0, sizeof wp);
    wp.width = w;
    wp.height = h;
    std::memcpy(wp.inv_view, cu + 32, sizeof wp.inv_view);
    std::memcpy(wp.inv_proj, cu + 48, sizeof wp.inv_proj);
    std::memcpy(wp.position, cu + 80, sizeof wp.position);
    wp.mass = pp[0];
    wp.spin = pp[1];
    const uint32_t fi = e->rt.frames; // paramsWithFrame.frameIndex = this.frameCount
    wp.jitter[0] = halton_f32((fi % 8u) + 1u, 2u) - 0.5f;
    wp.jitter[1] = halton_f32((fi % 8u) + 1u, 3u) - 0.5f;
    wp.max_steps = max_steps > 0 ? max_steps : 150;
    wp.tile_world = 1;
    wp.arith = arith;
    wp.stars = 1;
    rc = grv_render_frame_wgsl(e, &wp, compute_tex, nullptr, nullptr, stream);
    if (rc != GRV_OK) return rc;
    GRV_HIP(e, launch_post_quantize(compute_tex, w * h, s)); // texture_storage_2d<rgba16float>
    // Pass 2: ATAA resolve, history ping-pong (renderer.ts:319-345, 385-395)
    const uint32_t hi = e->rt.hist, nx = 1u - hi;
    AtaaCameraHost cam;
    std::memcpy(cam.inv_view, cu + 32, sizeof cam.inv_view);
    std::memcpy(cam.inv_proj, cu + 48, sizeof cam.inv_proj);
    std::memcpy(cam.prev_view_proj, cu + 64, sizeof cam.prev_view_proj);
    std::memcpy(cam.position, cu + 80, sizeof cam.position);
    const bool fast_post = arith != GRV_ARITH_STRICT; // the post chain follows the march's contract
    GRV_HIP(e, (fast_post ? launch_ataa_resolve_fast : launch_ataa_resolve)(w, h, cam, compute_tex, hist[hi], 1,
                                                                             hist[nx], s));
    // Pass 3: blit with Reinhard (renderer.ts:14-50, 397-411)
    GRV_HIP(e, (fast_post ? launch_blit_reinhard_fast : launch_blit_reinhard)(w, h, hist[nx], d_screen, s));
    e->rt.hist = nx;
    e->rt.frames++;
    return GRV_OK;
}

int grv_webgl_render(grv_engine *e, const GrvGlslParams *p, int32_t bloom_enabled, int32_t camera_moving,
                     float *d_screen, void *stream) {
    if (!e) return GRV_ERR_INVALID;
    if (!p || !d_screen) return fail(e, GRV_ERR_INVALID, "null argument");
    const uint32_t w = p->width, h = p->height;
    if (w == 0 || h == 0 || (uint64_t)w * h > (1ull << 27)) return fail(e, GRV_ERR_INVALID, "bad resolution");
    if (p->tile_world > 1) return fail(e, GRV_ERR_INVALID, "the renderer layer draws whole frames");
    hipStream_t s = static_cast<hipStream_t>(stream);
    GRV_HIP(e, hipSetDevice(e->device));
    int rc = ensure_targets(e, w, h, s);
    if (rc != GRV_OK) return rc;
    rc = ensure_bloom_scratch(e, w, h, s);
    if (rc != GRV_OK) return rc;
    const size_t plane = (size_t)w * h * 4;
    float *scene = e->rt.mem, *ping = e->rt.mem + plane, *pong = e->rt.mem + 2 * plane;
    // scene pass into the RGBA16F scene target, linear output (manager.ts:84-86)
    GrvGlslParams gp = *p;
    gp.tone_map = 0;
    rc = grv_render_frame_glsl(e, &gp, scene, nullptr, nullptr, stream);
    if (rc != GRV_OK) return rc;
    GRV_HIP(e, launch_post_quantize(scene, w * h, s));
    // ReprojectionManager.resolve: write index 0 -> write pong, read ping (reprojection.ts:209-216)
    float *write_tex = e->rt.hist == 0 ? pong : ping;
    const float *read_tex = e->rt.hist == 0 ? ping : pong;
    const bool fast_post = p->arith == GRV_ARITH_FAST; // the post chain follows the march's contract
    GRV_HIP(e, (fast_post ? launch_taa_resolve_fast : launch_taa_resolve)(w, h, scene, read_tex, 0.75f,
                                                                         camera_moving, 1, write_tex, s));
    e->rt.hist = 1u - e->rt.hist;
    // bloom (features.bloom) or plain presentation: both are the combine pass (bloom.ts:443-632)
    float *scratch = static_cast<float *>(e->post_mem);
    const auto bloom_fn = fast_post ? launch_bloom_fast : launch_bloom;
    if (bloom_enabled) {
        GRV_HIP(e, bloom_fn(w, h, write_tex, 0.8f, 0.5f, 2, 1, scratch, d_screen, s));
    } else {
        // drawTextureToScreen: combine with intensity 0; the bloom input is a stale dummy upstream,
        // here the (zero or last) bright-pass target, multiplied by 0 either way
        GRV_HIP(e, bloom_fn(w, h, write_tex, 3.0e38f, 0.0f, 0, 1, scratch, d_screen, s));
    }
    e->rt.frames++;
    return GRV_OK;
}

int grv_webgpu_render_host(grv_engine *e, const float *cu, const float *pp, int32_t max_steps,
                           int32_t arith, float *screen) {
    if (!e) return GRV_ERR_INVALID;
    if (!cu || !pp || !screen) return fail(e, GRV_ERR_INVALID, "null argument");
    const size_t bytes = (size_t)(uint32_t)pp[2] * (uint32_t)pp[3] * 4 * sizeof(float);
    if (bytes == 0 || bytes > ((size_t)1 << 31)) return fail(e, GRV_ERR_INVALID, "bad resolution");
    GRV_HIP(e, hipSetDevice(e->device));
    int rc = ensure_stage(e, bytes);
    if (rc != GRV_OK) return rc;
    rc = grv_webgpu_render(e, cu, pp, max_steps, arith, static_cast<float *>(e->stage_mem), nullptr);
    if (rc != GRV_OK) return rc;
    GRV_HIP(e, hipDeviceSynchronize());
    GRV_HIP(e, hipMemcpy(screen, e->stage_mem, bytes, hipMemcpyDeviceToHost));
    return GRV_OK;
}

int grv_webgl_render_host(grv_engine *e, const GrvGlslParams *p, int32_t bloom_enabled,
                          int32_t camera_moving, float *screen) {
    if (!e) return GRV_ERR_INVALID;
    if (!p || !screen) return fail(e, GRV_ERR_INVALID, "null argument");
    const size_t bytes = (size_t)p->width * p->height * 4 * sizeof(float);
    if (bytes == 0 || bytes > ((size_t)1 << 31)) return fail(e, GRV_ERR_INVALID, "bad resolution");
    GRV_HIP(e, hipSetDevice(e->device));
    int rc = ensure_stage(e, bytes);
    if (rc != GRV_OK) return rc;
    rc = grv_webgl_render(e, p, bloom_enabled, camera_moving, static_cast<float *>(e->stage_mem), nullptr);
    if (rc != GRV_OK) return rc;
    GRV_HIP(e, hipDeviceSynchronize());
    GRV_HIP(e, hipMemcpy(screen, e->stage_mem, bytes, hipMemcpyDeviceToHost));
    return GRV_OK;
}

} // extern "C"
