/* Type declarations of the addon, shaped like the .d.ts wasm-pack emits for
 * `#[wasm_bindgen] impl PhysicsEngine` (physics-engine/gravitas-wasm/src/lib.rs:56-465), plus the
 * frame surfaces this engine adds.  Numbers are f64 on the Rust side unless noted. */

export interface InitOutput {
  /** one module-wide arena; `new Float32Array(memory.buffer, engine.get_sab_ptr(), 2048)` works as
   *  with WebAssembly.Memory (src/workers/physics.worker.ts:61-68) */
  readonly memory: { readonly buffer: ArrayBuffer };
}

export default function init(): Promise<InitOutput>;
export function init_hooks(): void;
/** ArrayBuffer over page-locked host memory: `new Float32Array(allocPinned(w*h*16))` passed as
 *  renderFrame({out}) receives the frame in one DMA and no copy is made on the JS side */
export function allocPinned(bytes: number): ArrayBuffer;

export interface RenderFrameOptions {
  width: number;
  height: number;
  /** camera position; looks at `target` (default origin) with `up` (default +y) */
  eye?: [number, number, number];
  target?: [number, number, number];
  up?: [number, number, number];
  /** vertical field of view in degrees (default 60, WebGPUCanvas.tsx:143-151) */
  fovY?: number;
  maxSteps?: number;
  tolerance?: number;
  /** 0 endpoints only, 1 thin-disk (T x g) Planck-LUT shading */
  shading?: number;
  arith?: "fast" | "strict";
  /** radial temperature profile of the disk: the shader's closed form (default) or the
   *  Page-Thorne table generate_disk_lut() returns (physics/disk.rs:175-201) */
  diskProfile?: "shortcut" | "pageThorne";
  /** tile the image plane over the first n HIP devices of the node: each renders its round-robin
   *  share of 64x64 tiles, one gather of finished tiles to device 0 (RCCL over xGMI) */
  devices?: number;
  /** the same assembly path with n ranks on device 0 (one-GPU hosts, tests) */
  virtualRanks?: number;
  /** what the gather of a multi-device frame carries: f32 pixels (default) or the compute pass's own
   *  rgba16float format (half the bytes; the image is the one-device frame rounded through binary16) */
  exchange?: "rgba32f" | "rgba16f";
  /** render into this array (width*height*4) instead of a fresh one; renderFrameAsync into allocPinned()
   *  memory receives the DMA directly (no staging copy), two calls in flight overlap copy and kernels */
  out?: Float32Array;
  /** keep the frame in HBM: the (synchronous, one-device) call queues the frame into a new DeviceImage and
   *  returns a QueuedFrame at once */
  keepOnDevice?: boolean;
  /** the same into an image the caller keeps (createImage): a frame loop alternates two of them */
  image?: DeviceImage;
}

export interface IntegrateBatchOptions {
  method?: "rkf45" | "rk4" | "symplectic";
  metric?: "ks" | "bl" | "schwarzschild";
  /** IntegrationOptions (geodesic/integrator.rs:24-47); defaults are IntegrationOptions::default */
  tolerance?: number;
  initialStep?: number;
  maxSteps?: number;
  escapeRadius?: number;
  renormalizeInterval?: number;
  stepSize?: number;
  arith?: "strict" | "fast";
  /** IntegrationOptions.record_path (integrator.rs:32): the result gains `paths`, `counts`, `maxPoints` */
  recordPath?: boolean;
  /** rows of `paths` kept per ray (default maxSteps + 1, the longest path there is) */
  maxPoints?: number;
}

/** per ray: Trajectory.final_state / steps_taken / termination / max_hamiltonian_drift (geodesic/mod.rs:150-161) */
export interface IntegrateBatchResult {
  states: Float64Array;
  steps: Uint32Array;
  term: Uint8Array;
  drift: Float64Array;
  /** with recordPath: Trajectory.path (geodesic/mod.rs:160) as [n][maxPoints][8] f64 rows -- ray i holds
   *  min(counts[i], maxPoints) states: the initial state as passed in, then the state after every step */
  paths?: Float64Array;
  /** 1 + steps_taken: the length of the reference's Vec<GeodesicState> */
  counts?: Uint32Array;
  maxPoints?: number;
}

export interface RenderFrameResult {
  /** linear RGBA f32, row-major, top row first */
  rgba: Float32Array;
  width: number;
  height: number;
  rays: number;
  acceptedSteps: number;
  launches: number;
  devices: number;
  /** how the finished tiles reached device 0: "rccl" (one send/recv group over xGMI), "peer_copy", or
   * "none" for a one-device frame.  A multi-device request whose RCCL cannot be opened throws. */
  transport: "rccl" | "peer_copy" | "none";
}

export interface WebGLFrameOptions {
  width: number;
  height: number;
  mass?: number;
  spin?: number;
  zoom?: number;
  mouse?: [number, number];
  time?: number;
  maxRaySteps?: number;
  /** ShaderManager #defines as bits (include/gravitas_abi.h GRV_GLSL_*) */
  features?: number;
  bloom?: number;
  cameraMoving?: number;
  fast?: number;
  /** present into a device image (queued; returns QueuedFrame) */
  image?: DeviceImage;
  keepOnDevice?: boolean;
}

export class PhysicsEngine {
  constructor(mass: number, spin: number);
  free(): void;

  update_params(mass: number, spin: number): void;
  compute_horizon(): number;
  compute_isco(): number;
  compute_photon_sphere(): number;
  compute_dilation(r: number): number;
  compute_g_factor(r: number, lambda: number): number;

  /** lib.rs:422-464; input shorter than 8 is echoed */
  integrate_ray_relativistic(
    initial_state: Float64Array | number[], steps: number, tolerance: number, use_kerr_schild: boolean,
  ): Float64Array;
  /** the name BASELINE's north star uses for the same entry */
  integratePhotonGeodesic(
    initial_state: Float64Array | number[], steps: number, tolerance: number, use_kerr_schild: boolean,
  ): Float64Array;
  /** extension: arithmetic contract of the two entries above on this engine -- "strict" (default: the
   *  reference-order bits) or "fast" (the same geodesic to rounding, <= 1e-5 relative, a third of the time) */
  set_ray_arith(contract: "strict" | "fast"): void;

  generate_disk_lut(): Float32Array;
  get_disk_lut_ptr(): number;
  generate_spectrum_lut(width: number, height: number, max_temp: number): Float32Array;

  get_sab_ptr(): number;
  /** lib.rs:74.  `ptr` is a byte offset into `memory.buffer` (4-byte aligned, room for 2048 floats):
   *  tick_sab then reads its controls from and publishes into that block.  Anything else throws a
   *  RangeError (a raw pointer has no other meaning in JS). */
  attach_sab(ptr: number): void;
  get_sab_layout(): number[];
  set_camera_state(px: number, py: number, pz: number, lx: number, ly: number, lz: number): void;
  set_auto_spin(enabled: boolean): void;
  tick_sab(dt_override: number): void;

  compute_shadow_curve(theta_obs: number, n_points: number): Float32Array;
  compute_shadow_radius(): number;
  compute_shadow_shift(theta_obs: number): Float32Array;
  compute_disk_flux(r: number): number;

  generate_embedding_mesh(r_min: number, r_max: number, n_radial: number, n_angular: number): Float32Array;
  generate_ergosphere_mesh(n_polar: number, n_azimuthal: number): Float32Array;
  compute_kretschner(r: number, theta: number): number;
  generate_curvature_field(r_min: number, r_max: number, n_radial: number, n_polar: number): Float32Array;
  compute_light_cone_tilt(r: number, theta: number): number;
  generate_tilt_field(r_min: number, r_max: number, n_radial: number, n_polar: number): Float32Array;
  compute_frame_drag_omega(r: number, theta: number): number;
  generate_frame_drag_field(r_min: number, r_max: number, n_radial: number, n_polar: number): Float32Array;
  compute_flamm_height(r: number): number;
  compute_proper_distance(r1: number, r2: number, n_steps: number): number;

  /** f64 RKF45 frame (pixel -> ray of compute.wgsl.ts:159-187) */
  renderFrame(options: RenderFrameOptions): RenderFrameResult | QueuedFrame;
  render_frame(options: RenderFrameOptions): RenderFrameResult | QueuedFrame;
  /** the same frame on the libuv pool (an engine handle of its own): the caller's loop is not held */
  renderFrameAsync(options: RenderFrameOptions): Promise<RenderFrameResult>;
  /** n independent integrate() calls (geodesic/mod.rs:180-253) in one launch; states = 8 n f64 */
  integrate_batch(states: Float64Array, options?: IntegrateBatchOptions): IntegrateBatchResult;
  integrateBatch(states: Float64Array, options?: IntegrateBatchOptions): IntegrateBatchResult;
  integrateBatchAsync(states: Float64Array, options?: IntegrateBatchOptions): Promise<IntegrateBatchResult>;
  /** WebGPURenderer.render with the 352-byte / 32-byte uniform blocks (src/types/webgpu.ts:67-116) */
  renderWebGPUFrame(
    cameraUniforms: Float32Array, physicsParams: Float32Array,
    options?: { maxSteps?: number; arith?: "fast" | "strict"; image?: DeviceImage; keepOnDevice?: boolean },
  ): Float32Array | QueuedFrame;
  /** WebGLRenderer.render's scene + TAA + bloom chain */
  renderWebGLFrame(options: WebGLFrameOptions): Float32Array | QueuedFrame;

  // ---- device-resident frames (include/gravitas_abi.h "device images") ----
  // renderFrame / renderShaderFrame / renderWebGLFrame / renderWebGPUFrame with {image} or {keepOnDevice: true}
  // return QueuedFrame at once: the frame stays in HBM as the renderers' textures do upstream
  // (webgpu/renderer.ts:280-411); pixels cross PCIe in DeviceImage.read / readAsync only.
  /** streamOf: write the new image on that image's compute stream (frames in queue order); without it the image
   *  gets a stream of its own (frames into different images run concurrently) */
  createImage(width: number, height: number, options?: { streamOf?: DeviceImage }): DeviceImage;
  /** one f32 march launch (GLSL fragment march, configs[1]; WGSL compute march, configs[3]) without the post chain */
  renderShaderFrame(options: ShaderFrameOptions): QueuedFrame | { rgba: Float32Array; width: number; height: number; acceptedSteps: number };
  readImage(image: DeviceImage, out?: Float32Array): Float32Array;
  /** BloomManager.applyBloomToTexture (bloom.ts:443-583) between two device images; returns `out` */
  postBloom(scene: DeviceImage, out: DeviceImage, options?: { intensity?: number; threshold?: number; blurPasses?: number; halfStorage?: number; fast?: boolean }): DeviceImage;
  /** ReprojectionManager.resolve (reprojection.ts:196-262); returns `out` */
  postTaa(current: DeviceImage, history: DeviceImage, out: DeviceImage, options?: { blendFactor?: number; cameraMoving?: boolean; halfStorage?: number; fast?: boolean }): DeviceImage;
  /** counters stay in HBM across frames; frameStats() after a loop returns the sums */
  statsAccumulate(on: boolean): void;
  frameStats(): FrameStats;
  frameStatsReset(): void;
  /** waits for everything queued on the engine's device */
  synchronize(): void;
}

export interface FrameStats {
  rays: number; acceptedSteps: number; rkfTries: number; crossings: number; maxDrift: number;
  launches: number; integrateMs: number; termCount: number[];
}
export interface QueuedFrame { image: DeviceImage; width: number; height: number; rays: number; queued: true }
export interface ShaderFrameOptions {
  kernel?: "glsl" | "wgsl"; width: number; height: number; maxSteps?: number; arith?: "fast" | "strict" | "packed";
  eye?: [number, number, number]; target?: [number, number, number]; up?: [number, number, number]; fovY?: number;
  mass?: number; spin?: number; zoom?: number; time?: number; features?: number; mouse?: [number, number];
  image?: DeviceImage; keepOnDevice?: boolean; out?: Float32Array;
}
/** W x H RGBA f32 in HBM with a stream of its own; made by createImage or a {keepOnDevice: true} call */
export class DeviceImage {
  readonly width: number;
  readonly height: number;
  readonly bytes: number;
  /** the one D2H, behind everything queued on the image (allocPinned memory: one DMA) */
  read(out?: Float32Array): Float32Array;
  /** queued at once, settles when the copy has landed; `out` must live in allocPinned() memory */
  readAsync(out: Float32Array): Promise<Float32Array>;
  /** counters of the frame that last wrote the image (waits for this image only) */
  stats(): FrameStats;
  wait(): DeviceImage;
  ready(): boolean;
  free(): void;
}
