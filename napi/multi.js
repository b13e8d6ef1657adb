// napi/multi.js <G> : renderFrame({devices: G}) -- the image plane over G real HIP devices through
// the addon (grv_engine_create_multi, RCCL gather) -- against the one-device frame of the same
// camera.  Prints one JSON line; tests/test_gpu_multi_real.py runs it where G devices exist.
const wasm = require("./shim/blackhole_physics.js");
const G = parseInt(process.argv[2] || "2", 10);
(async () => {
  await wasm.default();
  const engine = new wasm.PhysicsEngine(1.0, 0.999);
  const same = (a, c) => a.length === c.length && a.every((v, i) => Object.is(v, c[i]));
  const res = { G, frames: [] };
  const eyes = [[59.55, -7.31, 0.0], [40.0, 12.0, 30.0], [10.0, 55.0, -20.0]];
  for (const arith of ["fast", "strict"]) {
    for (const eye of eyes) {
      const o = { width: 320, height: 200, eye, arith };
      const one = engine.renderFrame(o);
      const many = engine.renderFrame({ ...o, devices: G });
      res.frames.push({ arith, equal: same(one.rgba, many.rgba), steps_equal: one.acceptedSteps === many.acceptedSteps,
                        devices: many.devices, transport: many.transport, rays: many.rays });
    }
  }
  const a = await engine.renderFrameAsync({ width: 320, height: 200, eye: eyes[0], arith: "fast", devices: G });
  const b = engine.renderFrame({ width: 320, height: 200, eye: eyes[0], arith: "fast" });
  res.async_equal = same(a.rgba, b.rgba) && a.devices === G;
  console.log(JSON.stringify(res));
  engine.free();
})().catch((e) => { console.error("FAILED", e); process.exit(1); });
