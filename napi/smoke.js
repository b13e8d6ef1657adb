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
  };
  console.log(JSON.stringify(res));
  engine.free();
})().catch((e) => { console.error("FAILED", e); process.exit(1); });
