// Runs on the GPU box: node napi/worker_protocol.js
// Walks the SharedArrayBuffer protocol the reference's worker and bridge speak
// (src/workers/physics.worker.ts:111-176 writer, src/engine/physics-bridge.ts:148-188 reader)
// against the addon: engine memory -> shared buffer between two sequence increments, seqlock read.
const path = require("path");
const wasm = require(path.join(__dirname, "blackhole_physics.node"));
const OFFSETS = { CONTROL: 0, CAMERA: 64, PHYSICS: 128, TELEMETRY: 256, LUTS: 2048 }; // lib.rs:36-40

(async () => {
  const mod = await wasm.default();
  const engine = new wasm.PhysicsEngine(1.0, 0.9);               // INIT{mass: 1, spin: 0.9}
  engine.set_auto_spin(true);
  engine.set_camera_state(3.0, 4.0, 12.0, 0, 0, 0);

  // main thread side: the 2 MiB shared buffer and its views (physics-bridge.ts:53-67, 97-123)
  const sab = new SharedArrayBuffer(2 * 1024 * 1024);
  const seqView = new Int32Array(sab);
  const cameraView = new Float32Array(sab, OFFSETS.CAMERA * 4, OFFSETS.PHYSICS - OFFSETS.CAMERA);
  const physicsView = new Float32Array(sab, OFFSETS.PHYSICS * 4, OFFSETS.TELEMETRY - OFFSETS.PHYSICS);

  // worker side: views onto engine memory, rebound from memory.buffer + get_sab_ptr()
  const wasmF32 = new Float32Array(mod.memory.buffer);
  const ticks = [];
  let lastSeen = -1, torn = 0;
  for (let k = 0; k < 4; k++) {
    const startIdx = engine.get_sab_ptr() / 4;
    wasmF32[startIdx + OFFSETS.CONTROL + 1] = 0.5 * k;           // mouse_dx
    wasmF32[startIdx + OFFSETS.CONTROL + 3] = -0.1;              // zoom delta
    engine.tick_sab(Math.min(0.016, 0.033));
    Atomics.add(seqView, OFFSETS.TELEMETRY, 1);                  // write started
    cameraView.set(wasmF32.subarray(startIdx + OFFSETS.CAMERA, startIdx + OFFSETS.PHYSICS));
    physicsView.set(wasmF32.subarray(startIdx + OFFSETS.PHYSICS, startIdx + OFFSETS.TELEMETRY));
    Atomics.add(seqView, OFFSETS.TELEMETRY, 1);                  // write complete

    // reader (bridge.tick): two loads around the read, NaN guard
    const seq1 = Atomics.load(seqView, OFFSETS.TELEMETRY);
    const cam = Array.from(cameraView.subarray(0, 12)), phys = Array.from(physicsView);
    const seq2 = Atomics.load(seqView, OFFSETS.TELEMETRY);
    if (seq1 !== seq2 || seq1 === lastSeen) torn++;
    lastSeen = seq1;
    ticks.push({ seq: seq1, finite: cam.slice(0, 3).every(Number.isFinite), camera: cam, physics: phys,
                 inputs_consumed: wasmF32[startIdx + 1] === 0 && wasmF32[startIdx + 3] === 0 });
  }
  // attach_sab (lib.rs:74): tick_sab moves to another block of memory.buffer; get_sab_ptr does not
  const own = engine.get_sab_ptr(), other = 512 * 1024;
  engine.attach_sab(other);
  wasmF32.fill(0, other / 4, other / 4 + 2048);
  wasmF32[other / 4 + OFFSETS.CONTROL + 3] = -0.1;
  const before = Array.from(wasmF32.subarray(own / 4 + OFFSETS.CAMERA, own / 4 + OFFSETS.CAMERA + 3));
  engine.tick_sab(0.016);
  const attach = { ptr_unchanged: engine.get_sab_ptr() === own,
                   published_there: wasmF32[other / 4 + OFFSETS.PHYSICS + 2] === 1.0 &&
                                    Number.isFinite(wasmF32[other / 4 + OFFSETS.CAMERA]),
                   control_consumed_there: wasmF32[other / 4 + OFFSETS.CONTROL + 3] === 0,
                   own_block_untouched: before.every((v, i) => v === wasmF32[own / 4 + OFFSETS.CAMERA + i]),
                   errors: [] };
  for (const bad of [-4, 2, 1 << 20, (1 << 20) - 4096, NaN, 1e12]) {
    try { engine.attach_sab(bad); attach.errors.push(null); } catch (e) { attach.errors.push(e instanceof RangeError); }
  }
  engine.attach_sab(own);
  console.log(JSON.stringify({ ticks: ticks, torn: torn, attach: attach }));
  engine.free();
})().catch((e) => { console.error("FAILED", e); process.exit(1); });
