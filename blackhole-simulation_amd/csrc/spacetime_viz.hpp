// spacetime_viz.hpp -- the spacetime read-outs of PhysicsEngine next to the path
// (gravitas-wasm/src/lib.rs:139-159, 214-306 over
// gravitas-core/src/spacetime/{curvature,lightcone,frame_drag,embedding}.rs).
// Scalars are evaluated on the host, grids and meshes by one thread per grid point.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace grvhip {

struct VizHole {
    double mass;
    double spin_raw; // as given to the engine (lib.rs:44-45)
    double a_bl;     // clamped spin * mass: the engine's Boyer-Lindquist metric (kerr.rs:48-74)
};

enum VizField : int { kVizKretschner = 0, kVizLightConeTilt = 1, kVizFrameDrag = 2 };

double viz_kretschner(const VizHole &bh, double r, double theta);       // curvature.rs:22-47
double viz_light_cone_tilt(const VizHole &bh, double r, double theta);  // lightcone.rs:19-48
double viz_frame_drag_omega(const VizHole &bh, double r, double theta); // kerr.rs:143-152
double viz_flamm_height(double r, double mass);                         // embedding.rs:17-23
double viz_proper_distance(const VizHole &bh, double r1, double r2, size_t n_steps); // :51-65

// (r, theta, value) f32 triples, row = radial index (curvature.rs:49-68 and siblings)
hipError_t launch_viz_field(int field, const VizHole &bh, double r_min, double r_max,
                            uint32_t n_radial, uint32_t n_polar, float *d_out, hipStream_t s);
// xyz f32 vertices (embedding.rs:70-111 ; frame_drag.rs:40-69)
hipError_t launch_embedding_mesh(const VizHole &bh, double r_min, double r_max, uint32_t n_radial,
                                 uint32_t n_angular, float *d_out, hipStream_t s);
hipError_t launch_ergosphere_mesh(const VizHole &bh, uint32_t n_polar, uint32_t n_azimuthal,
                                  float *d_out, hipStream_t s);

} // namespace grvhip
