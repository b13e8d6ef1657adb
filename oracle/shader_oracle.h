/*
 * shader_oracle.h -- CPU (f32) restatement of the reference's two GPU march loops:
 *   WGSL compute kernel   src/shaders/compute.wgsl.ts:28-258      
 *   GLSL fragment march   src/shaders/blackhole/fragment.glsl.ts:40-221,
 *                         chunks/metric.ts:13-149 (a16), chunks/disk.ts:16-115 (a17),
 *                         chunks/blackbody.ts:9-34
 * TEST INFRASTRUCTURE ONLY.  The reference holds no image or pixel test and its
 * noise textures are unseeded Math.random(), so whole frames have no reference-held answer
 * ("parity unpinned" for pixels).  What IS pinned (tests/test_f32_oracle_pins.py, through the
 * orc_hook_* entry points below, none of it through the HIP twin): get_derivatives and
 * symplectic_step against the pinned f64 oracle (compute.wgsl.ts:42-120 == kerr.rs:412-499),
 * horizon / ISCO / photon sphere against kerr.rs:507-554, the disk Doppler factor against
 * redshift.rs:65-95, the shader expressions the reference tests itself
 * (src/__tests__/physics/advanced-physics.test.ts), kerr_geodesic_accel against its closed form,
 * and both marches against the critical impact parameter of their own equation of motion
 * (independent DOP853 integration).
 */
#ifndef SHADER_ORACLE_H
#define SHADER_ORACLE_H

#include <stddef.h>
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
    int32_t stars;      /* 1: the escape-branch star hash of compute.wgsl.ts:201-204; 0: skipped */
} orc_wgsl_params;

/* one pixel; rgba[4]; returns the number of symplectic steps taken */
uint32_t orc_wgsl_pixel(const orc_wgsl_params *p, uint32_t ix, uint32_t iy, float rgba[4]);
void orc_wgsl_frame(const orc_wgsl_params *p, uint32_t stride_x, uint32_t stride_y, float *rgba,
                    uint32_t *steps, int nthreads);
/* wgsl_f64_twin.c: the same march evaluated in double (analysis instrument for disagreements between
 * two f32 forms of it).  exit class: 0 horizon, 1 escape, 2 step budget, 3 opaque */
uint32_t orc_wgsl_pixel_f64(const orc_wgsl_params *p, uint32_t ix, uint32_t iy, double rgb[3],
                            int32_t *exit_class, double *min_r);
void orc_wgsl_pixels_f64(const orc_wgsl_params *p, size_t n, const uint32_t *xy, double *rgb, uint32_t *steps,
                         int32_t *cls, double *min_r, double *axis_margin, int nthreads);

/* ShaderManager #defines (src/shaders/manager.ts:61-82) as bits */
#define ORC_GLSL_LENSING 1u
#define ORC_GLSL_DISK 2u
#define ORC_GLSL_DOPPLER 4u
#define ORC_GLSL_STARS 8u
#define ORC_GLSL_PHOTON_GLOW 16u
#define ORC_GLSL_JETS 32u       /* only effective with DISK (manager.ts:72-73) */
#define ORC_GLSL_REDSHIFT 64u
#define ORC_GLSL_DITHER 128u    /* blue-noise start offset (fragment.glsl.ts:104-108); always on upstream */

/* uniforms of the GLSL fragment shader (chunks/common.ts:8-35) */
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
    float time;              /* u_time */
    float turbulence;        /* >= 0: stands in for noise()*0.5 + noise()*0.25 of disk.ts:55;
                                < 0: the two noise() fetches from the noise texture */
    int32_t max_ray_steps;   /* u_maxRaySteps, clamped to 500 in the shader */
    int32_t tone_map;        /* 0 = ENABLE_LINEAR_OUTPUT, 1 = ACES + gamma */
    uint32_t features;       /* ORC_GLSL_* */
    int32_t quality;         /* 0 = RAY_QUALITY_OFF/LOW indicator path (fragment.glsl.ts:76-88) */
    float show_redshift, show_kerr_shadow, debug; /* u_show_redshift, u_show_kerr_shadow, u_debug */
    float cam_pos[3], cam_quat[4];                /* u_camPos, u_camQuat (xyzw) */
    float shadow_count;                           /* u_shadowCount */
    float shadow_curve[64][2];                    /* u_shadowCurve */
    const uint8_t *noise_r;  /* 256x256 R channel of u_noiseTex (LINEAR, REPEAT) */
    const uint8_t *blue_r;   /* 256x256 R channel of u_blueNoiseTex (NEAREST, REPEAT) */
} orc_glsl_params;

/* deterministic stand-in for createNoiseTexture's Math.random() bytes
 * (src/utils/webgl-utils.ts:259-305: floor(random * 255), RGBA8, 256x256): xorshift32 stream */
void orc_seeded_noise_rgba8(uint32_t seed, uint32_t size, uint8_t *rgba);

uint32_t orc_glsl_pixel(const orc_glsl_params *p, uint32_t ix, uint32_t iy, float rgba[4]);
void orc_glsl_frame(const orc_glsl_params *p, uint32_t stride_x, uint32_t stride_y, float *rgba,
                    uint32_t *steps, int nthreads);

/* test hooks (see the end of shader_oracle.c): single building blocks of the two marches */
void orc_hook_wgsl_derivs(const float x[4], const float p[4], float M, float spin, float dx[4], float dp[4]);
void orc_hook_wgsl_step(const float x[4], const float p[4], float h, float M, float spin, float ox[4], float op[4]);
float orc_hook_wgsl_horizon(float M, float a);
float orc_hook_wgsl_isco(float M, float a);
void orc_hook_glsl_accel(const float p[3], const float v[3], float M, float a, float acc[3], float *omega);
float orc_hook_glsl_horizon(float M, float a);
float orc_hook_glsl_isco(float M, float a);
float orc_hook_glsl_photon_sphere(float M, float a);
float orc_hook_glsl_redshift_potential(float rs, float r);
float orc_hook_glsl_ergosphere_radius(float M, float a, float cos_theta);
float orc_hook_glsl_beaming(float delta);
float orc_hook_glsl_disk_delta(float M, float a, float spin, float r, float L_photon);
void orc_hook_glsl_blackbody(float temp, float rgb[3]);
void orc_hook_glsl_verlet_oscillator(float *x, float *v, float k, float dt, int steps);

#ifdef __cplusplus
}
#endif
#endif
