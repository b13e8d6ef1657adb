/*
 * frame_oracle.h -- CPU statement of the frame path: pixel -> initial state ->
 * integrate -> thin-disk redshift shading.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference has NO CPU render-frame: frames exist only as
 * WebGL/WebGPU shaders.  This file therefore DEFINES the f64 frame as
 *   pixel->state   = src/shaders/compute.wgsl.ts:159-187 evaluated in f64,
 *   integration    = gravitas-core integrate() (gravitas_oracle.c),
 *   disk crossing  = the theta = pi/2 sign test of compute.wgsl.ts:217, with the
 *                    crossing radius linearly interpolated inside the step,
 *   g-factor       = physics/redshift.rs:65-95 (kerr_g_factor),
 *   temperature    = src/shaders/blackhole/chunks/disk.ts:100-102 profile,
 *   colour         = bilinear lookup in physics/spectrum.rs:76-102's (T x g) LUT.
 * "parity unpinned": no reference test or artefact holds a pixel value (F6/F7).
 */
#ifndef FRAME_ORACLE_H
#define FRAME_ORACLE_H

#include "gravitas_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Camera block: f64 mirror of the fields of CameraUniforms the compute kernel
 * reads (src/shaders/types.wgsl.ts:6-17: inv_view, inv_proj, position);
 * matrices column-major like gl-matrix / WGSL mat4x4. */
typedef struct {
    double position[3];
    double inv_view[16];
    double inv_proj[16];
    double pixel_offset[2]; /* uv = (id + offset) / size ; (0.5,0.5) = pixel centres */
} orc_camera;

typedef struct {
    uint32_t width, height;
    int metric_kind;
    double mass, spin;
    orc_options opt;
    int shading;            /* 0: endpoints only, 1: thin-disk LUT shading */
    double disk_inner;      /* <= 0: prograde ISCO */
    double disk_outer;      /* compute.wgsl.ts:217 uses 30 */
    double disk_temp;       /* K, src/configs/simulation.config.ts:158-160 default 9500 */
    double disk_opacity;    /* alpha added per crossing (compute.wgsl.ts:233 uses 0.6*T) */
    double exposure;        /* multiplies LUT rgb */
    uint32_t lut_width, lut_height;
    double lut_max_temp;
    int disk_profile;       /* 0: the shader's closed form (disk.ts:100-102); 1: the 512-entry
                               Page-Thorne table of generate_temperature_lut (physics/disk.rs:175-201)
                               read with linear interpolation, clamped (the LINEAR / CLAMP_TO_EDGE
                               texture of src/rendering/webgl/renderer.ts:436-446) */
    const float *disk_lut;  /* 512 entries for disk_profile 1; NULL: orc_render_frame generates it */
} orc_frame_params;

typedef struct {
    uint64_t rays;
    uint64_t accepted_steps;
    uint64_t rkf_tries;
    uint64_t term_count[5];
    uint64_t crossings;
    double max_drift;
} orc_frame_stats;

/* gl-matrix style helpers (src/components/canvas/WebGPUCanvas.tsx:143-157) */
void orc_camera_look_at(const double eye[3], const double target[3], const double up[3],
                        double fovy_rad, double aspect, orc_camera *cam);

/* compute.wgsl.ts:159-187 in f64 */
void orc_pixel_state(const orc_camera *cam, uint32_t width, uint32_t height, uint32_t i,
                     uint32_t j, orc_state *out);

/* one pixel: returns termination; rgba may be NULL; rcross[2] may be NULL */
int orc_trace_pixel(const orc_camera *cam, const orc_frame_params *fp, const float *lut,
                    uint32_t i, uint32_t j, orc_state *final_state, uint32_t *steps,
                    uint32_t *tries, double *drift, uint32_t *ncross, double *rcross,
                    float rgba[4]);

/* whole frame or a strided subset (pixel (i,j) traced iff i%stride_x==0 && j%stride_y==0;
 * outputs are dense over the subset, row-major).  Any output pointer may be NULL.
 * lut may be NULL (generated internally when shading). */
void orc_render_frame(const orc_camera *cam, const orc_frame_params *fp, const float *lut,
                      uint32_t stride_x, uint32_t stride_y, float *rgba, orc_state *final_states,
                      uint32_t *steps, uint8_t *term, double *drift, orc_frame_stats *stats,
                      int nthreads);

/* bilinear (T, g) lookup used by the shading rule */
void orc_lut_sample(const float *lut, uint32_t w, uint32_t h, double max_temp, double temp,
                    double g, double rgb[3]);

/* temperature profile, src/shaders/blackhole/chunks/disk.ts:100-102 in f64 */
double orc_disk_temp_profile(double r, double disk_inner);
/* Page-Thorne profile: linear read of the normalised table over [rin, rout] (width entries) */
double orc_disk_lut_profile(const float *lut, uint32_t width, double r, double rin, double rout);

#ifdef __cplusplus
}
#endif
#endif
