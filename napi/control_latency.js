// What the worker's control-plane calls cost while frames are in flight: node napi/control_latency.js [outfile]
// The reference's worker (src/workers/physics.worker.ts:75-176) ticks every ~13 ms and, on a parameter change, asks
// for LUTs, shadow curves and meshes; with this engine the SAME device may have tens of milliseconds of frame kernels
// queued.  Each call is timed on an idle device and with three 4K frames (81 ms of work) queued behind it.
"use strict";
const path = require("path");
const fs = require("fs");
const wasm = require(path.join(__dirname, "blackhole_physics.node"));
const nowMs = () => Number(process.hrtime.bigint()) / 1e6;

(async () => {
  await wasm.default();
  const engine = new wasm.PhysicsEngine(1.0, 0.999);
  const W = 3840, H = 2160, th = 97 * Math.PI / 180, eye = [60 * Math.sin(th), 60 * Math.cos(th), 0];
  const imgs = [engine.createImage(W, H), engine.createImage(W, H)];
  const frame = (k) => engine.renderFrame({ width: W, height: H, eye: eye, arith: "fast", tolerance: 1e-8, image: imgs[k % 2] });
  const calls = {
    tick_sab: () => engine.tick_sab(0.016),
    compute_horizon: () => engine.compute_horizon(),
    compute_shadow_curve_64: () => engine.compute_shadow_curve(1.2, 64),
    generate_disk_lut: () => engine.generate_disk_lut(),
    generate_spectrum_lut_512x64: () => engine.generate_spectrum_lut(512, 64, 1e5),
    generate_embedding_mesh_64x64: () => engine.generate_embedding_mesh(2.0, 30.0, 64, 64),
    integrate_ray_relativistic: () => engine.integrate_ray_relativistic(new Float64Array([0, 20, Math.PI / 2, 0, -1, -1, 0, 3.5]), 2000, 1e-8, true),
    integrate_batch_256: () => engine.integrate_batch(new Float64Array(8 * 256).map((_, i) => [0, 20 + (i >> 3) * 0.1, 1.5, 0, -1, -1, 0, 3.5][i & 7]), { maxSteps: 2000 }),
  };
  frame(0); engine.synchronize();
  for (const f of Object.values(calls)) f();   // warm: first-use allocations and code loads are not what is measured
  engine.synchronize();
  const res = { frame_ms: null, calls: {} };
  let t0 = nowMs(); frame(0); engine.synchronize(); res.frame_ms = +(nowMs() - t0).toFixed(3);
  for (const [name, f] of Object.entries(calls)) {
    const idle = [], loaded = [];
    for (let rep = 0; rep < 5; rep++) {
      engine.synchronize();
      t0 = nowMs(); f(); idle.push(nowMs() - t0);
      engine.synchronize();
      frame(0); frame(1); frame(2);          // ~81 ms of kernels queued, nothing waited for
      t0 = nowMs(); f(); loaded.push(nowMs() - t0);
    }
    engine.synchronize();
    const med = (a) => a.slice().sort((x, y) => x - y)[a.length >> 1];
    res.calls[name] = { idle_ms: +med(idle).toFixed(3), under_three_queued_4k_frames_ms: +med(loaded).toFixed(3) };
  }
  const out = JSON.stringify(res);
  if (process.argv[2]) fs.writeFileSync(process.argv[2], out + "\n");
  console.log(out);
  engine.free();
})().catch((e) => { console.error("FAILED", e); process.exit(1); });
