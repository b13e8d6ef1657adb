/*
 * post_oracle.h -- f32 CPU restatement of the reference's post chain:
 *   TAA resolve (WebGL)   src/shaders/postprocess/reprojection.glsl.ts:44-116
 *                         driven by src/rendering/reprojection.ts:196-262
 *   ATAA resolve (WebGPU) src/shaders/postprocess/ataa.wgsl.ts:29-86
 *   bloom                 src/shaders/postprocess/bloom.glsl.ts:35-127,
 *                         pass sequence src/rendering/bloom.ts:443-583 (renderScale = 1)
 * TEST INFRASTRUCTURE ONLY.  The reference holds no test that executes these shaders
 * (its pipeline tests mock WebGL): parity unpinned upstream; pinned here
 * by closed-form properties (tests/test_post_chain.py) and, without going through the HIP twin,
 * by hand-computed 3x3 / constant / impulse images evaluated in numpy f64 from the shaders'
 * literal constants (tests/test_f32_oracle_pins.py: 1.5 sigma / 2 sigma clip boxes, the
 * variance weight, feedback 0.92, luminance coefficients, the 9-tap weights, ACES, gamma).
 *
 * Images are RGBA f32, row-major.  Texture fetches are GL LINEAR + CLAMP_TO_EDGE with f32
 * weights; render targets are RGBA16F upstream, modelled by rounding every stored channel to
 * binary16 (round-to-nearest-even) when `half_storage` is set.
 */
#ifndef POST_ORACLE_H
#define POST_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

float orc_round_to_half(float x);

/* reprojection.glsl.ts: out = YCoCg-clipped history blend (1.5 sigma box, variance weight) */
void orc_taa_resolve(uint32_t w, uint32_t h, const float *current, const float *history,
                     float blend_factor, int camera_moving, int half_storage, float *out);

typedef struct {
    float inv_view[16], inv_proj[16], prev_view_proj[16]; /* column-major, types.wgsl.ts:6-17 */
    float position[3];
} orc_ataa_camera;
/* ataa.wgsl.ts: 2 sigma box, history re-projected at depth 12, feedback 0.92 */
void orc_ataa_resolve(uint32_t w, uint32_t h, const orc_ataa_camera *cam, const float *current,
                      const float *history, int half_storage, float *out);

/* bloom.ts:443-583: bright pass (w/2 x h/2) -> blur_passes x (H, V) at w/4 x h/4 -> combine
 * (scene + bloom * intensity, ACES, gamma) at w x h.  scratch-free: allocates internally. */
void orc_bloom(uint32_t w, uint32_t h, const float *scene, float threshold, float intensity,
               int blur_passes, int half_storage, float *out);

#ifdef __cplusplus
}
#endif
#endif
