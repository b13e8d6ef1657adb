// Runs on the GPU box: node napi/bulk.js [outfile]
// The bulk entries of the addon: integrate_batch (4096 geodesics in one call), the *Async forms
// (libuv pool, the caller's loop keeps ticking), renderFrame({virtualRanks / devices, out}) and
// allocPinned.  Prints one JSON line; tests/test_napi_addon.py holds it to the oracle.
const path = require("path");
const wasm = require(path.join(__dirname, "blackhole_physics.node"));

function rays(n) { // deterministic fan of photons from r = 20 .. 60 towards the hole
  const s = new Float64Array(8 * n);
  for (let i = 0; i < n; i++) {
    const u = (i + 0.5) / n;
    s.set([0, 20 + 40 * ((i * 7919) % n) / n, Math.PI / 2 - 0.4 + 0.8 * u, 0.1 * i, -1, -1, 0.3 - 0.6 * u,
           -8 + 16 * u], 8 * i);
  }
  return s;
}
const now = () => { const t = process.hrtime(); return t[0] * 1e3 + t[1] / 1e6; };

(async () => {
  await wasm.default();
  const engine = new wasm.PhysicsEngine(1.0, 0.9);
  const n = 4096, init = rays(n);
  const opts = { maxSteps: 2000, tolerance: 1e-8 };
  const res = {};
  // one call, 4096 rays
  let t0 = now();
  const b = engine.integrate_batch(init, opts);
  res.batch_first_ms = now() - t0;
  t0 = now();
  for (let k = 0; k < 5; k++) engine.integrate_batch(init, opts);
  res.batch_ms = (now() - t0) / 5;
  res.batch = { n: n, init: Array.from(init.slice(0, 64)), states: Array.from(b.states), steps: Array.from(b.steps),
                term: Array.from(b.term), drift: Array.from(b.drift) };
  // the same rays one per call through the FFI entry (first 32)
  t0 = now();
  const single = [];
  for (let i = 0; i < 32; i++)
    single.push(Array.from(engine.integrate_ray_relativistic(init.subarray(8 * i, 8 * i + 8), 2000, 1e-8, true)));
  res.single_ms_per_ray = (now() - t0) / 32;
  res.single = single;
  res.us_per_ray_batch = res.batch_ms * 1e3 / n;
  // async forms: the loop keeps running while the GPU works
  let ticks = 0;
  const timer = setInterval(() => { ticks++; engine.tick_sab(0.016); }, 1);
  const pa = engine.integrateBatchAsync(init, opts);
  const pf = engine.renderFrameAsync({ width: 640, height: 360, eye: [59.55, -7.31, 0.0], arith: "fast" });
  const pf2 = engine.renderFrameAsync({ width: 96, height: 54, eye: [59.55, -7.31, 0.0], virtualRanks: 3 });
  const [ba, fa, fa2] = await Promise.all([pa, pf, pf2]);
  clearInterval(timer);
  res.async = { ticks_during: ticks,
                batch_equal: ba.states.every((v, i) => Object.is(v, b.states[i])) && ba.steps.every((v, i) => v === b.steps[i]),
                frame_rays: fa.rays, frame_steps: fa.acceptedSteps };
  // synchronous frame: one device, virtual ranks, caller-owned pinned output
  const f1 = engine.renderFrame({ width: 96, height: 54, eye: [59.55, -7.31, 0.0] });
  const f4 = engine.renderFrame({ width: 96, height: 54, eye: [59.55, -7.31, 0.0], virtualRanks: 4 });
  const pinned = new Float32Array(wasm.allocPinned(96 * 54 * 16));
  const fp = engine.renderFrame({ width: 96, height: 54, eye: [59.55, -7.31, 0.0], devices: 1, out: pinned });
  const same = (a, c) => a.length === c.length && a.every((v, i) => Object.is(v, c[i]));
  // the exchange in the compute pass's own rgba16float format: the one-device frame rounded through binary16
  const f4h = engine.renderFrame({ width: 96, height: 54, eye: [59.55, -7.31, 0.0], virtualRanks: 4, exchange: "rgba16f" });
  const hb = new Float32Array(1), hu = new Uint32Array(hb.buffer);
  const toHalf = (x) => {  // round to nearest even through binary16; every step below is exact in doubles
    hb[0] = x;
    const bits = hu[0], neg = bits >>> 31, e = (bits >>> 23) & 255;
    if (e === 255) return x;
    let val;
    if (e - 127 > 15) val = Infinity;
    else if (e - 127 >= -14) {  // normal half: keep ten mantissa bits
      let m = bits & 0x7fffff;
      const rem = m & 0x1fff, lsb = (m >>> 13) & 1;
      m = (m >>> 13) + ((rem > 0x1000 || (rem === 0x1000 && lsb)) ? 1 : 0);
      val = (1 + m / 1024) * Math.pow(2, e - 127);
      if (val > 65504) val = Infinity;
    } else {  // subnormal half: a multiple of 2^-24
      const q = Math.abs(x) * Math.pow(2, 24);
      let k = Math.floor(q);
      const d = q - k;
      if (d > 0.5 || (d === 0.5 && (k & 1))) k += 1;
      val = k * Math.pow(2, -24);
    }
    return neg ? -val : val;
  };
  const f4back = engine.renderFrame({ width: 96, height: 54, eye: [59.55, -7.31, 0.0], virtualRanks: 4 });
  res.frames_half = { equal_rounded: f4h.rgba.every((v, i) => Object.is(v, Math.fround(toHalf(f1.rgba[i])))),
                      differs_from_f32: !f4h.rgba.every((v, i) => Object.is(v, f1.rgba[i])),
                      back_to_f32: f4back.rgba.every((v, i) => Object.is(v, f1.rgba[i])) };
  res.frames = { steps1: f1.acceptedSteps, steps4: f4.acceptedSteps, devices4: f4.devices,
                 ranks_equal: same(f1.rgba, f4.rgba), async_ranks_equal: same(f1.rgba, fa2.rgba),
                 pinned_equal: same(f1.rgba, pinned), pinned_is_out: fp.rgba === pinned || fp.rgba.buffer === pinned.buffer };
  // update_params reaches the multi handle; bad requests throw
  engine.update_params(1.0, 0.5);
  const g1 = engine.renderFrame({ width: 64, height: 36, eye: [59.55, -7.31, 0.0] });
  const g2 = engine.renderFrame({ width: 64, height: 36, eye: [59.55, -7.31, 0.0], virtualRanks: 4 });
  res.frames.update_reaches_ranks = same(g1.rgba, g2.rgba);
  const errs = [];
  for (const f of [() => engine.integrate_batch(new Float64Array(7)), () => engine.renderFrame({ width: 8, height: 8, devices: 99 }),
                   () => engine.renderFrame({ width: 8, height: 8, out: new Float32Array(3) })]) {
    try { f(); errs.push(null); } catch (e) { errs.push(String(e.message)); }
  }
  let rejected = null;
  try { await engine.renderFrameAsync({ width: 8, height: 8, devices: 63 }); } catch (e) { rejected = String(e.message); }
  res.errors = { sync: errs, async_rejected: rejected };
  res.empty = engine.integrate_batch(new Float64Array(0)).steps.length;
  // recordPath: Trajectory.path of the first 16 rays (engine is at spin 0.5 here), 64 rows per ray
  const pp = engine.integrate_batch(init.slice(0, 128), { maxSteps: 2000, tolerance: 1e-8, recordPath: true, maxPoints: 64 });
  res.paths = { init: Array.from(init.slice(0, 128)), counts: Array.from(pp.counts), maxPoints: pp.maxPoints,
                rows: Array.from(pp.paths), steps: Array.from(pp.steps) };
  // memory JS can reach while a work is queued: the async forms work on their own copies
  {
    const { MessageChannel } = require("worker_threads");
    const o = { width: 96, height: 54, eye: [59.55, -7.31, 0.0] };
    engine.update_params(1.0, 0.9);
    const want = engine.renderFrame(o).rgba;
    const keepOut = new Float32Array(96 * 54 * 4), goneOut = new Float32Array(96 * 54 * 4);
    const pk = engine.renderFrameAsync({ ...o, out: keepOut });
    const pg = engine.renderFrameAsync({ ...o, out: goneOut });
    const ch = new MessageChannel();
    ch.port1.postMessage(goneOut.buffer, [goneOut.buffer]);   // detached while the frame is queued
    const st2 = init.slice(0, 512);
    const pb = engine.integrateBatchAsync(st2, opts);
    st2.fill(NaN);                                            // overwritten while the batch is queued
    let detachedRejected = null;
    try { await pg; } catch (e) { detachedRejected = String(e.message); }
    const fk = await pk, bb = await pb;
    ch.port1.close(); ch.port2.close();
    const ref = engine.integrate_batch(init.slice(0, 512), opts);
    res.async_memory = { out_filled: same(want, keepOut) && fk.rgba === keepOut, detached_rejected: detachedRejected,
                         input_copy: bb.states.every((v, i) => Object.is(v, ref.states[i])) };
    engine.update_params(1.0, 0.5);
  }
  // free() with a work still queued: the promise still settles
  const e2 = new wasm.PhysicsEngine(1.0, 0.7);
  const late = e2.integrateBatchAsync(init.slice(0, 800), opts);
  e2.free();
  res.free_while_pending = (await late).steps.length;
  const out = JSON.stringify(res);
  if (process.argv[2]) require("fs").writeFileSync(process.argv[2], out + "\n");
  console.log(out);
  engine.free();
})().catch((e) => { console.error("FAILED", e); process.exit(1); });
