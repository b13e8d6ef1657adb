// BASELINE's frame loops driven from the host north_star names: JavaScript through the N-API addon.
//
//   node napi/bench_frames.js [--config c3|c2|c4|c5] [--form device|read|async|host|all] [--steps K] [--warmup W]
//                             [--in-flight 1|2] [--eye R0,THETA_DEG] [--out file.jsonl]
//
// Same workloads, cameras, warm-up and timed window as bench.py (which calls the same C ABI through
// Python ctypes): c3 = BASELINE configs[2], 3840x2160, a = 0.999, RKF45 tol 1e-8, <= 2048 steps, f64 FAST +
// Planck LUT, camera at r0 = 60 M, theta = 97 deg; c2 = configs[1], 1920x1080, GLSL Verlet march, 512 steps,
// default preset, two frames in flight.  One JSON line per form:
//   device : renderFrame({image}) / renderShaderFrame({image}) -- the frame stays in HBM (a DeviceImage); the
//            loop only queues, as the reference's renderer only submits (webgpu/renderer.ts:280-411); what
//            bench.py's `value` measures (outputs resident in HBM)
//   read   : the same + image.readAsync(pinned) per frame: frame i's D2H under frame i+1's kernels
//   async  : renderFrameAsync({out: pinned}) with two frames in flight (c3 only: the f64 frame's async form)
//   host   : renderFrame() / renderShaderFrame() as the reference's consumers call it: a new Float32Array
//            per frame, synchronous
// `value` = accepted ray-steps of the K timed frames / wall time, M ray-steps/s (BASELINE.json's metric).
"use strict";
const path = require("path");
const fs = require("fs");
const wasm = require(path.join(__dirname, "blackhole_physics.node"));

function parseArgs(argv) {
  const a = { config: "c3", form: "all", steps: null, warmup: null, inFlight: null, eye: null, out: null,
              width: 0, height: 0, readImages: 2 };
  for (let i = 2; i < argv.length; i++) {
    const k = argv[i], v = argv[i + 1];
    if (k === "--config") { a.config = v; i++; }
    else if (k === "--form") { a.form = v; i++; }
    else if (k === "--steps") { a.steps = parseInt(v, 10); i++; }
    else if (k === "--warmup") { a.warmup = parseInt(v, 10); i++; }
    else if (k === "--in-flight") { a.inFlight = parseInt(v, 10); i++; }
    else if (k === "--eye") { a.eye = v.split(",").map(Number); i++; }
    else if (k === "--out") { a.out = v; i++; }
    else if (k === "--read-images") { a.readImages = parseInt(v, 10); i++; }
    else if (k === "--width") { a.width = parseInt(v, 10); i++; }
    else if (k === "--height") { a.height = parseInt(v, 10); i++; }
    else throw new Error("unknown argument " + k);
  }
  if (["c2", "c3", "c4", "c5"].indexOf(a.config) < 0) throw new Error("--config c2|c3|c4|c5");
  if (a.steps === null) a.steps = a.config === "c2" ? 300 : 20;
  if (a.warmup === null) a.warmup = a.config === "c2" ? 60 : 3;
  if (a.inFlight === null) a.inFlight = a.config === "c2" ? 2 : 1;  // bench.py's defaults
  return a;
}
const nowMs = () => Number(process.hrtime.bigint()) / 1e6;

async function main() {
  const args = parseArgs(process.argv);
  await wasm.default();
  const cfg = args.config;
  const c3 = cfg === "c3" || cfg === "c5";   // the f64 frame (c5: tol 1e-9 under the reference-order STRICT contract)
  const W = args.width || { c2: 1920, c3: 3840, c4: 7680, c5: 3840 }[cfg], H = args.height || { c2: 1080, c3: 2160, c4: 4320, c5: 2160 }[cfg];
  const r0 = args.eye ? args.eye[0] : 60.0, thDeg = args.eye ? args.eye[1] : 97.0;
  const th = thDeg * Math.PI / 180;
  const eye = [r0 * Math.sin(th), r0 * Math.cos(th), 0.0];
  const engine = new wasm.PhysicsEngine(1.0, 0.999);
  const frameOpts = cfg === "c3" ? { width: W, height: H, eye: eye, arith: "fast", tolerance: 1e-8, maxSteps: 2048 }
    : cfg === "c5" ? { width: W, height: H, eye: eye, arith: "strict", tolerance: 1e-9, maxSteps: 2048 }
    : cfg === "c4" ? { width: W, height: H, kernel: "wgsl", arith: "packed", maxSteps: 1024, eye: eye }
    : { width: W, height: H, kernel: "glsl", arith: "fast", maxSteps: 512 };
  const render = (extra) => c3 ? engine.renderFrame(Object.assign({}, frameOpts, extra))
                               : engine.renderShaderFrame(Object.assign({}, frameOpts, extra));
  const workload = cfg === "c3"
    ? W + "x" + H + " frame, a=0.999 Kerr-Schild, adaptive RKF45 (tol 1e-08) <= 2048 steps, f64 FAST + Planck LUT (T x g) redshift shading"
    : cfg === "c5" ? W + "x" + H + " frame, a=0.999 Kerr-Schild, adaptive RKF45 (tol 1e-09) <= 2048 steps, f64 STRICT (reference order)"
    : cfg === "c4" ? W + "x" + H + " frame, a=0.999, f32 WGSL compute march, fixed 1024-step budget, two rays per lane"
    : W + "x" + H + " frame, a=0.999, GLSL fragment Verlet march <= 512 steps, default preset, f32 FAST";
  const lines = [];
  const emit = (form, elapsedMs, steps, frames, extra) => {
    const line = Object.assign({
      metric: "Mray-steps/s", value: +(steps / (elapsedMs / 1e3) / 1e6).toFixed(2), unit: "Mray-steps/s", n_gpus: 1,
      steps: frames, warmup: args.warmup, ms_per_step: +(elapsedMs / frames).toFixed(4), higher_is_better: true,
      dtype: c3 ? "f64" : "f32", data: "synthetic",
      host: "node " + process.version + " through napi/blackhole_physics.node (N-API over the C ABI)",
      form: form,
      config: { workload: workload, baseline_config: { c2: "configs[1]", c3: "configs[2]", c4: "configs[3]", c5: "configs[4]" }[cfg], arith: frameOpts.arith,
                eye: { r0: r0, theta_deg: thDeg }, rays: W * H, accepted_steps_per_frame: Math.round(steps / frames),
                frames_in_flight: args.inFlight },
    }, extra || {});
    lines.push(line);
    console.log(JSON.stringify(line));
  };
  const want = (f) => args.form === "all" || args.form === f;
  const K = args.steps, Wm = args.warmup, nfl = args.inFlight;

  // ---- (c) device-resident: the loop only queues ----
  if (want("device")) {
    const imgs = [];
    for (let k = 0; k < nfl; k++) imgs.push(engine.createImage(W, H));
    engine.statsAccumulate(true);  // counters stay in HBM across frames: no read-back in the loop
    for (let i = 0; i < Wm; i++) render({ image: imgs[i % nfl] });
    engine.synchronize();
    engine.frameStatsReset();
    engine.synchronize();
    const t0 = nowMs();
    for (let i = 0; i < K; i++) render({ image: imgs[(Wm + i) % nfl] });
    const tq = nowMs();
    engine.synchronize();
    const t1 = nowMs();
    const st = engine.frameStats();
    engine.statsAccumulate(false);
    emit("device", t1 - t0, st.acceptedSteps, K,
         { host_queue_ms_per_frame: +((tq - t0) / K).toFixed(4), host_waits_in_frame_loop: 0,
           pixels: "stay in HBM (DeviceImage); no D2H in the timed region" });
    imgs.forEach((im) => im.free());
  }

  // ---- (b') device image + one asynchronous D2H per frame into pinned memory ----
  if (want("read")) {
    // `nfl` compute streams (bench.py's frames in flight), two images in rotation on each: image k is written on
    // stream k % nfl in queue order and read back on its own copy stream
    const n = args.readImages * nfl, imgs = [], outs = [];
    for (let k = 0; k < n; k++) {
      imgs.push(k < nfl ? engine.createImage(W, H) : engine.createImage(W, H, { streamOf: imgs[k % nfl] }));
      outs.push(new Float32Array(wasm.allocPinned(W * H * 16)));
    }
    engine.statsAccumulate(true);
    const pend = new Array(n).fill(null);
    const step = async (i) => {
      const k = i % n;
      if (pend[k]) { await pend[k]; pend[k] = null; }     // image k / buffer k are free again
      render({ image: imgs[k] });
      pend[k] = imgs[k].readAsync(outs[k]);               // queued behind the frame on the image's stream
    };
    for (let i = 0; i < Wm; i++) await step(i);
    await Promise.all(pend.filter(Boolean)); pend.fill(null);
    engine.synchronize();
    engine.frameStatsReset();
    engine.synchronize();
    const t0 = nowMs();
    for (let i = 0; i < K; i++) await step(Wm + i);
    await Promise.all(pend.filter(Boolean));
    const t1 = nowMs();
    const st = engine.frameStats();
    engine.statsAccumulate(false);
    emit("read", t1 - t0, st.acceptedSteps, K,
         { pixels: "every frame copied to page-locked host memory (image.readAsync), frame i's D2H under frame i+1's kernels",
           d2h_bytes_per_frame: W * H * 16, images_in_rotation: n, compute_streams: nfl });
    imgs.forEach((im) => im.free());
  }

  // ---- (b) renderFrameAsync into two pinned `out` buffers ----
  if (want("async") && cfg === "c3") {
    const outs = [new Float32Array(wasm.allocPinned(W * H * 16)), new Float32Array(wasm.allocPinned(W * H * 16))];
    const pend = [null, null];
    let steps = 0;
    const step = async (i, count) => {
      const k = i % 2;
      if (pend[k]) { const r = await pend[k]; pend[k] = null; if (r.count) steps += r.res.acceptedSteps; }
      pend[k] = engine.renderFrameAsync(Object.assign({}, frameOpts, { out: outs[k] })).then((res) => ({ res: res, count: count }));
    };
    const drain = async () => {
      for (let k = 0; k < 2; k++) if (pend[k]) { const r = await pend[k]; pend[k] = null; if (r.count) steps += r.res.acceptedSteps; }
    };
    for (let i = 0; i < Wm; i++) await step(i, false);
    await drain();
    const t0 = nowMs();
    for (let i = 0; i < K; i++) await step(Wm + i, true);
    await drain();
    const t1 = nowMs();
    emit("async", t1 - t0, steps, K,
         { pixels: "renderFrameAsync({out: allocPinned}) x 2 in flight: the DMA lands in the caller's array, no staging copy",
           d2h_bytes_per_frame: W * H * 16 });
  }

  // ---- (a) the synchronous host form, as the consumers call it ----
  if (want("host")) {
    const Kh = Math.min(K, c3 ? 10 : 100), Wh = Math.min(Wm, 3);
    let steps = 0;
    for (let i = 0; i < Wh; i++) render({});
    const t0 = nowMs();
    for (let i = 0; i < Kh; i++) steps += render({}).acceptedSteps;
    const t1 = nowMs();
    emit("host", t1 - t0, steps, Kh,
         { pixels: "a new Float32Array per frame (pageable), synchronous call", d2h_bytes_per_frame: W * H * 16 });
  }
  engine.free();
  if (args.out) fs.writeFileSync(args.out, lines.map((l) => JSON.stringify(l)).join("\n") + "\n");
}
main().catch((e) => { console.error("FAILED", e); process.exit(1); });
