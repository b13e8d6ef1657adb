// Replacement for the wasm-pack output `public/wasm/blackhole_physics.js` of the reference
// (package.json:10 builds it with `wasm-pack --target web --out-name blackhole_physics`; the module
// alias "blackhole-physics" in tsconfig.json:22 / vitest.config.ts:13-16 points at it).
// It re-exports the N-API addon, which presents the same surface:
//   default export  init()  -> Promise<{ memory: { buffer: ArrayBuffer } }>
//   class PhysicsEngine, function init_hooks
// Consumers (src/engine/physics-bridge.ts, src/workers/physics.worker.ts) stay unchanged.
import { createRequire } from "node:module";

const require = createRequire(import.meta.url);
// adjust the relative path to where the addon was built (make -C napi)
const addon = require(process.env.BLACKHOLE_PHYSICS_ADDON || "../blackhole_physics.node");

export default addon.default;
export const PhysicsEngine = addon.PhysicsEngine;
export const init_hooks = addon.init_hooks;
// extensions of this engine (device-resident frames, page-locked host memory)
export const DeviceImage = addon.DeviceImage;
export const allocPinned = addon.allocPinned;
