// Runs on the GPU box: node napi/frames_check.js <outfile> [W H]
// Full-size frames through every form the addon offers, reduced to SHA-256 digests of the pixel bytes and the
// accepted-step totals; tests/test_napi_frames.py renders the same frames through Python ctypes (the path
// bench.py times) and requires equal digests -- i.e. bit-equal pixels -- and equal step totals.
"use strict";
const path = require("path");
const crypto = require("crypto");
const fs = require("fs");
const wasm = require(path.join(__dirname, "blackhole_physics.node"));
const sha = (f32) => crypto.createHash("sha256").update(Buffer.from(f32.buffer, f32.byteOffset, f32.byteLength)).digest("hex");

(async () => {
  await wasm.default();
  const W = parseInt(process.argv[3] || "3840", 10), H = parseInt(process.argv[4] || "2160", 10);
  const th = 97 * Math.PI / 180, eye = [60 * Math.sin(th), 60 * Math.cos(th), 0.0];
  const engine = new wasm.PhysicsEngine(1.0, 0.999);
  const o = { width: W, height: H, eye: eye, arith: "fast", tolerance: 1e-8, maxSteps: 2048 };
  const res = { W: W, H: H };
  // (c) device-resident: queued, then one explicit read into page-locked memory
  const pinned = new Float32Array(wasm.allocPinned(W * H * 16));
  const q = engine.renderFrame(Object.assign({ keepOnDevice: true }, o));
  res.device = { queued: q.queued === true, has_rgba: "rgba" in q, w: q.image.width, h: q.image.height, bytes: q.image.bytes };
  const st = q.image.stats();
  res.device.steps = st.acceptedSteps;
  res.device.rays = st.rays;
  engine.readImage(q.image, pinned);
  res.device.sha = sha(pinned);
  // the same image again (caller-kept), read into a plain array
  engine.renderFrame(Object.assign({ image: q.image }, o));
  res.device.sha_again = sha(q.image.read());
  // readAsync under a ticking loop
  let ticks = 0;
  const timer = setInterval(() => { ticks++; }, 1);
  engine.renderFrame(Object.assign({ image: q.image }, o));
  pinned.fill(0);
  const back = await q.image.readAsync(pinned);
  clearInterval(timer);
  res.device.read_async = { sha: sha(pinned), same_array: back === pinned, ticks: ticks };
  // (b) renderFrameAsync into two pinned buffers, two in flight: no staging copy (the result IS the caller's array)
  const outs = [new Float32Array(wasm.allocPinned(W * H * 16)), new Float32Array(wasm.allocPinned(W * H * 16))];
  const [a0, a1] = await Promise.all([engine.renderFrameAsync(Object.assign({ out: outs[0] }, o)),
                                      engine.renderFrameAsync(Object.assign({ out: outs[1] }, o))]);
  res.async = { sha0: sha(outs[0]), sha1: sha(outs[1]), steps0: a0.acceptedSteps, steps1: a1.acceptedSteps,
                is_out: a0.rgba === outs[0] && a1.rgba === outs[1] };
  // async into an ordinary array still goes through staging and fills it
  const plain = new Float32Array(W * H * 4);
  const a2 = await engine.renderFrameAsync(Object.assign({ out: plain }, o));
  res.async.plain_sha = sha(plain);
  res.async.plain_steps = a2.acceptedSteps;
  // (a) the synchronous host form
  const f = engine.renderFrame(o);
  res.host = { sha: sha(f.rgba), steps: f.acceptedSteps };
  // c2: the GLSL march into an image, and the WGSL march (packed) -- 1080p
  const g = engine.renderShaderFrame({ width: 1920, height: 1080, kernel: "glsl", arith: "fast", maxSteps: 512, keepOnDevice: true });
  res.glsl = { steps: g.image.stats().acceptedSteps, sha: sha(g.image.read()) };
  const gh = engine.renderShaderFrame({ width: 1920, height: 1080, kernel: "glsl", arith: "fast", maxSteps: 512 });
  res.glsl.host_sha = sha(gh.rgba);
  res.glsl.host_steps = gh.acceptedSteps;
  const wg = engine.renderShaderFrame({ width: 1920, height: 1080, kernel: "wgsl", arith: "packed", maxSteps: 512, eye: eye, keepOnDevice: true });
  res.wgsl = { steps: wg.image.stats().acceptedSteps, sha: sha(wg.image.read()) };
  // post chain between images: bloom of the GLSL frame, TAA of two frames
  const bl = engine.createImage(1920, 1080);
  engine.postBloom(g.image, bl, { fast: true });
  res.bloom = { sha: sha(bl.read()) };
  const g2 = engine.renderShaderFrame({ width: 1920, height: 1080, kernel: "glsl", arith: "fast", maxSteps: 512, time: 0.5, keepOnDevice: true });
  const taa = engine.createImage(1920, 1080);
  engine.postTaa(g2.image, g.image, taa, { fast: true });
  res.taa = { sha: sha(taa.read()) };
  // the WebGL renderer presenting into two alternating images (history carried inside the engine)
  const scr = [engine.createImage(640, 360), engine.createImage(640, 360)];
  const shas = [];
  for (let i = 0; i < 4; i++) engine.renderWebGLFrame({ width: 640, height: 360, spin: 0.9, time: 0.1 * i, fast: 1, image: scr[i % 2] });
  shas.push(sha(scr[0].read()), sha(scr[1].read()));
  res.webgl = { sha_frame2: shas[0], sha_frame3: shas[1] };
  // errors
  const errs = [];
  for (const fn of [() => engine.renderFrame(Object.assign({ image: scr[0] }, o)),
                    () => engine.renderFrame(Object.assign({ image: {} }, o)),
                    () => scr[0].readAsync(new Float32Array(640 * 360 * 4)),
                    () => engine.postBloom(scr[0], scr[0])]) {
    try { fn(); errs.push(null); } catch (e) { errs.push(String(e.message)); }
  }
  let rej = null;
  try { await engine.renderFrameAsync(Object.assign({ keepOnDevice: true }, o)); } catch (e) { rej = String(e.message); }
  res.errors = { sync: errs, async_keep: rej };
  // free(): an image outlives its engine; a freed image says so
  const keep = engine.createImage(64, 36);
  engine.renderFrame({ width: 64, height: 36, eye: eye, arith: "fast", image: keep });
  scr[1].free();
  let freedMsg = null;
  try { scr[1].read(); } catch (e) { freedMsg = String(e.message); }
  engine.free();
  res.lifetime = { read_after_engine_free: keep.read().length, freed_msg: freedMsg };
  const out = JSON.stringify(res);
  if (process.argv[2]) fs.writeFileSync(process.argv[2], out + "\n");
  console.log(out);
})().catch((e) => { console.error("FAILED", e); process.exit(1); });
