/*
 * shader_oracle.h -- CPU (f32) restatement of the reference's two GPU march loops:
 *   WGSL compute kernel   src/shaders/compute.wgsl.ts:28-258      
 *   GLSL fragment march   src/shaders/blackhole/fragment.glsl.ts:40-221,
 *                         chunks/metric.ts:13-149 (a16), chunks/disk.ts:16-115 (a17),
 *                         chunks/blackbody.ts:9-34
 * TEST INFRASTRUCTURE ONLY.  "parity unpinned": the reference holds no image or
 * pixel test and its noise textures are unseeded Math.random();
 * see the .c file for what is restated and what is fixed to a constant.
 */
#ifndef SHADER_ORACLE_H
#define SHADER_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CameraUniforms + PhysicsParams fields the WGSL kernel reads (types.wgsl.ts:6-30) */
typedef struct {
    float inv_view[16];
    float inv_proj[16];
    float position[3];
    float mass, spin;
    uint32_t width, height;
    float jitter[2];    /* halton(frame)-0.5, in pixels (compute.wgsl.ts:154-157) */
    int32_t max_steps;  /* override MAX_STEPS (compute.wgsl.ts:13) */
} orc_wgsl_params;

/* one pixel; rgba[4]; returns the number of symplectic steps taken */
uint32_t orc_wgsl_pixel(const orc_wgsl_params *p, uint32_t ix, uint32_t iy, float rgba[4]);
void orc_wgsl_frame(const orc_wgsl_params *p, uint32_t stride_x, uint32_t stride_y, float *rgba,
                    uint32_t *steps, int nthreads);

/* uniforms of the GLSL fragment shader (chunks/common.ts:8-35) that the march uses */
typedef struct {
    uint32_t width, height;
    float mass;              /* u_mass */
    float spin;              /* u_spin as uploaded = spin * mass (webgl/renderer.ts:326) */
    float zoom;              /* u_zoom = zoom * 2 (renderer.ts:327) */
    float mouse[2];          /* u_mouse */
    float disk_size;         /* u_disk_size */
    float disk_scale_height; /* u_disk_scale_height */
    float disk_density;      /* u_disk_density */
    float disk_temp;         /* u_disk_temp */
    float lensing_strength;  /* u_lensing_strength */
    float time;              /* u_time (only rotates the turbulence phase) */
    float turbulence;        /* value standing in for noise()*0.5 + noise()*0.25 (unseeded texture) */
    int32_t max_ray_steps;   /* u_maxRaySteps, clamped to 500 in the shader */
    int32_t tone_map;        /* 0 = ENABLE_LINEAR_OUTPUT, 1 = ACES + gamma */
} orc_glsl_params;

uint32_t orc_glsl_pixel(const orc_glsl_params *p, uint32_t ix, uint32_t iy, float rgba[4]);
void orc_glsl_frame(const orc_glsl_params *p, uint32_t stride_x, uint32_t stride_y, float *rgba,
                    uint32_t *steps, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
