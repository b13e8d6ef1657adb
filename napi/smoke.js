// Runs on the GPU box: node napi/smoke.js  (Node >= 10; the image has v12).
// Exercises the wasm-bindgen-shaped surface the reference's TypeScript consumes
// (src/engine/physics-bridge.ts, src/workers/physics.worker.ts).
const path = require("path");
const wasm = require(path.join(__dirname, "blackhole_physics.node"));
(async () => {
  const mod = await wasm.default();                       // physics-bridge.ts:87-88
  const engine = new wasm.PhysicsEngine(1.0, 0.9);        // physics.worker.ts:64
  const sab = new Float32Array(mod.memory.buffer, engine.get_sab_ptr(), 2048); // worker :68
  engine.set_auto_spin(false);
  engine.tick_sab(0.016);
  const out = engine.integrate_ray_relativistic(
    new Float64Array([0, 20, Math.PI / 2, 0, -1, -1, 0, 3.5]), 10000, 1e-8, true);
  const echo = engine.integratePhotonGeodesic(new Float64Array([1, 2, 3]), 10, 1e-8, true);
  const res = {
    horizon: engine.compute_horizon(), isco: engine.compute_isco(),
    layout: engine.get_sab_layout(), sab_horizon: sab[128], sab_isco: sab[129], sab_points: sab[143],
    sab_seq: sab[256], ray: Array.from(out), echo: Array.from(echo),
    lut0: Array.from(engine.generate_spectrum_lut(8, 2, 1e5).slice(28, 32)),
    disk_lut_len: engine.generate_disk_lut().length,
    shadow_pts: engine.compute_shadow_curve(Math.PI / 2, 32).length / 2,
    shadow_shift: Array.from(engine.compute_shadow_shift(Math.PI / 2)),
    disk_lut_max: Math.max.apply(null, Array.from(
      new Float32Array(mod.memory.buffer, engine.get_disk_lut_ptr(), 512))),
    embedding_len: engine.generate_embedding_mesh(2.5, 30, 24, 16).length,   // physics-bridge.ts:315
    ergosphere_len: engine.generate_ergosphere_mesh(17, 12).length,          // physics-bridge.ts:333
    kretschner: engine.compute_kretschner(6.0, Math.PI / 2),
    tilt: engine.compute_light_cone_tilt(6.0, Math.PI / 2),
    omega: engine.compute_frame_drag_omega(6.0, Math.PI / 2),
    flamm: engine.compute_flamm_height(100.0),
    proper: engine.compute_proper_distance(4.0, 20.0, 500),
    curvature_len: engine.generate_curvature_field(2.2, 40, 8, 5).length,
    tilt_len: engine.generate_tilt_field(2.2, 40, 8, 5).length,
    drag_len: engine.generate_frame_drag_field(2.2, 40, 8, 5).length,
  };
  // the frame surface: f64 RKF45 kernel behind renderFrame (SURVEY F2)
  const frame = engine.renderFrame({ width: 96, height: 54, eye: [59.55, -7.31, 0.0], maxSteps: 2048 });
  let lit = 0;
  for (let i = 0; i < frame.rgba.length; i += 4) if (frame.rgba[i] + frame.rgba[i + 1] + frame.rgba[i + 2] > 0) lit++;
  res.frame = { rays: frame.rays, acceptedSteps: frame.acceptedSteps, lit: lit, alpha0: frame.rgba[3] };
  // the one-ray entry under the FAST contract, and back
  engine.set_ray_arith("fast");
  res.ray_fast = Array.from(engine.integratePhotonGeodesic(
    new Float64Array([0, 20, Math.PI / 2, 0, -1, -1, 0, 3.5]), 10000, 1e-8, true));
  engine.set_ray_arith("strict");
  res.ray_again = Array.from(engine.integrate_ray_relativistic(
    new Float64Array([0, 20, Math.PI / 2, 0, -1, -1, 0, 3.5]), 10000, 1e-8, true));
  try { engine.set_ray_arith("sloppy"); res.bad_arith = "accepted"; } catch (e) { res.bad_arith = e.constructor.name; }
  // the renderers' frame surfaces (WebGLRenderer.render / WebGPURenderer.render)
  const gl = engine.renderWebGLFrame({ width: 64, height: 36, spin: 0.9, maxRaySteps: 200 });
  let glMax = 0;
  for (let i = 0; i < gl.length; i += 4) glMax = Math.max(glMax, gl[i], gl[i + 1], gl[i + 2]);
  res.webgl = { len: gl.length, max: glMax, alpha: gl[3] };
  // arena slots: hot reload / repeated construction reuses the slots of freed engines, and running
  // out of slots is an exception, never another engine's region
  const firstPtr = engine.get_sab_ptr();
  const seen = new Set();
  for (let k = 0; k < 300; k++) {
    const e2 = new wasm.PhysicsEngine(1.0, 0.1);
    seen.add(e2.get_sab_ptr());
    e2.free();
  }
  const live = [];
  let exhausted = null;
  try {
    for (let k = 0; k < 200; k++) live.push(new wasm.PhysicsEngine(1.0, 0.2));
  } catch (e) {
    exhausted = String(e.message);
  }
  const ptrs = new Set(live.map((e2) => e2.get_sab_ptr()));
  res.arena = { reusedSlots: seen.size, reuseAvoidsLive: !seen.has(firstPtr), liveEngines: live.length,
                distinctLivePtrs: ptrs.size, liveAvoidFirst: !ptrs.has(firstPtr), exhausted: exhausted };
  live.forEach((e2) => e2.free());
  console.log(JSON.stringify(res));
  engine.free();
})().catch((e) => { console.error("FAILED", e); process.exit(1); });
