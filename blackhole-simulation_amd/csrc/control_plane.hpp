// control_plane.hpp -- the small pieces of PhysicsEngine next to the path
// (SURVEY.md 8f-1/2): Novikov-Thorne disk flux / temperature LUT, Bardeen shadow
// curve, camera filter, tick_sab.  Declarations shared by engine.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace grvhip {

// Page-Thorne flux at one radius (host scalar; gravitas-core/src/physics/disk.rs:90-151)
double page_thorne_flux_host(double r, double mass, double spin_clamped, double m_dot);
// 512-entry normalised temperature LUT on the GPU (disk.rs:175-201): d_out[width] floats,
// d_scratch[width] doubles
hipError_t launch_disk_temperature_lut(float *d_out, double *d_scratch, uint32_t width, double mass,
                                       double spin_clamped, hipStream_t s);

// Bardeen critical curve (gravitas-core/src/physics/shadow.rs:81-183): (alpha, beta) pairs
std::vector<double> bardeen_shadow_host(double mass, double spin_clamped, double theta_obs,
                                        size_t n_points);
double schwarzschild_shadow_radius_host(double mass);

struct CameraFilter { // gravitas-wasm/src/camera.rs:10-39
    double position[3] = {0.0, 0.0, 20.0};
    double velocity[3] = {0.0, 0.0, 0.0};
    double orientation[4] = {0.0, 1.0, 0.0, 0.0};
    bool auto_spin = false;
    bool finite() const;
    void update(double mouse_dx, double mouse_dy, double zoom_delta, double dt); // camera.rs:42-70
};

// tick_sab (gravitas-wasm/src/lib.rs:308-409) on a 2048-float block
void tick_sab_host(float *sab, double mass, double spin, double spin_clamped, double horizon,
                   double isco, CameraFilter &cam, CameraFilter &last_good, double dt_override);

} // namespace grvhip
