"""MI355X-native Kerr geodesic ray-marching engine (HIP, gfx950).

Host-side mirror of the reference's physics FFI
(physics-engine/gravitas-wasm/src/lib.rs:56-465, `PhysicsEngine`) over the C ABI
of include/gravitas_abi.h.  All compute runs in libgravitas_hip.so on the GPU;
there is no CPU fallback in this package.
"""
from .engine import (  # noqa: F401
    ARITH_FAST, ARITH_FAST_PACKED, ARITH_STRICT, DISK_PROFILE_PAGE_THORNE, DISK_PROFILE_SHORTCUT, KERR_BL, KERR_KS, METHOD_RK4, METHOD_RKF45, METHOD_SYMPLECTIC,
    SCHWARZSCHILD, TERM_DISK_CROSSING, TERM_ESCAPE, TERM_HORIZON, TERM_MAXSTEPS, TERM_NONE,
    Camera, FrameBuffers, FrameStats, GlslParams, GravitasError, Options, PhysicsEngine,
    RenderParams, WgslParams, build_library, camera_look_at, glsl_params, library_path,
    load_library, render_params, seeded_noise_rgba8, unpack_tiles, wgsl_params,
    GLSL_LENSING, GLSL_DISK, GLSL_DOPPLER, GLSL_STARS, GLSL_PHOTON_GLOW, GLSL_JETS, GLSL_REDSHIFT,
    GLSL_DITHER, GLSL_FEATURES_DEFAULT, AtaaParams, BloomParams, TaaParams,
    MultiEngine, TRANSPORT_AUTO, TRANSPORT_PEER_COPY, TRANSPORT_RCCL, rccl_probe,
    EXCHANGE_RGBA16F, EXCHANGE_RGBA32F, unlock_test_hooks, DeviceImage, SCHEDULE_DEFAULT, SCHEDULE_SLOT_ORDER, FAULT_NONE, FAULT_RENDER, FAULT_SEND, FAULT_PEER_COPY,
)
