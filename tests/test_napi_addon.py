"""The N-API addon (napi/gravitas_napi.c) presents the wasm-bindgen surface of
gravitas-wasm/src/lib.rs:56-465 to the reference's TypeScript (src/engine/physics-bridge.ts,
src/workers/physics.worker.ts).  CPU: it loads in Node and exports every method the
reference's FFI has; GPU: napi/smoke.js walks the worker's call sequence and its numbers are
checked against the oracle."""
import json
import math
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADDON = os.path.join(ROOT, "napi", "blackhole_physics.node")
NODE = shutil.which("node")

# every `pub fn` of `#[wasm_bindgen] impl PhysicsEngine` (lib.rs:56-465); attach_sab's raw pointer is
# a byte offset into `memory.buffer` (the arena), anything else throws
WASM_METHODS = [
    "attach_sab",
    "update_params", "compute_horizon", "compute_isco", "compute_photon_sphere", "compute_dilation",
    "generate_disk_lut", "get_disk_lut_ptr", "get_sab_ptr", "set_camera_state", "set_auto_spin",
    "generate_spectrum_lut", "generate_embedding_mesh", "generate_ergosphere_mesh",
    "compute_shadow_curve", "compute_shadow_radius", "compute_shadow_shift", "compute_disk_flux",
    "compute_g_factor", "compute_kretschner", "generate_curvature_field", "compute_light_cone_tilt",
    "generate_tilt_field", "compute_frame_drag_omega", "generate_frame_drag_field",
    "compute_flamm_height", "compute_proper_distance", "tick_sab", "get_sab_layout",
    "integrate_ray_relativistic",
    # names BASELINE.json's north_star uses for the path
    "integratePhotonGeodesic", "renderFrame", "free",
    # the renderers' frame surfaces
    "renderWebGPUFrame", "renderWebGLFrame",
    # bulk / non-blocking entries
    "integrate_batch", "integrateBatch", "integrateBatchAsync", "renderFrameAsync",
    # contract of the one-ray entry (grv_engine_set_ray_arith)
    "set_ray_arith",
    # device-resident frames (ABI 8 device images): VERDICT r5 item 1
    "createImage", "renderShaderFrame", "readImage", "postBloom", "postTaa", "statsAccumulate", "frameStats",
    "frameStatsReset", "synchronize",
]

pytestmark = pytest.mark.skipif(NODE is None or not os.path.exists(ADDON),
                                reason="node or the built addon is not available")


def _node(script, timeout=120):
    return subprocess.run([NODE, "-e", script], capture_output=True, text=True, timeout=timeout,
                          cwd=ROOT)


def test_addon_loads_and_exports_the_wasm_bindgen_surface():
    r = _node("const m=require(%r);console.log(JSON.stringify({k:Object.keys(m),"
              "p:Object.getOwnPropertyNames(m.PhysicsEngine.prototype),d:typeof m.default}))" % ADDON)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout)
    assert {"PhysicsEngine", "DeviceImage", "default", "init_hooks", "allocPinned"} <= set(got["k"]) and got["d"] == "function"
    missing = [m for m in WASM_METHODS if m not in got["p"]]
    assert not missing, missing


def test_esm_shim_reexports_the_addon(tmp_path):
    """napi/shim/blackhole_physics.js is what replaces public/wasm/blackhole_physics.js: the
    consumers' `import init, { PhysicsEngine } from "blackhole-physics"` must keep working."""
    script = tmp_path / "t.mjs"
    script.write_text(
        'import init, { PhysicsEngine, init_hooks } from "%s";\n'
        'init().then((mod) => console.log(JSON.stringify({i: typeof init, c: typeof PhysicsEngine,'
        ' h: typeof init_hooks, m: mod.memory.buffer.byteLength})));\n'
        % os.path.join(ROOT, "napi", "shim", "blackhole_physics.js"))
    r = subprocess.run([NODE, str(script)], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["i"] == got["c"] == got["h"] == "function" and got["m"] >= 2048 * 4


def test_type_declarations_match_the_addon():
    """napi/shim/blackhole_physics.d.ts (what wasm-pack would emit) declares what the addon exports."""
    dts = open(os.path.join(ROOT, "napi", "shim", "blackhole_physics.d.ts")).read()
    for name in ("PhysicsEngine", "DeviceImage"):
        cls = dts[dts.index("export class %s" % name):]
        cls = cls[:cls.index("\n}\n") + 3]
        declared = set(re.findall(r"^  (\w+)\(", cls, flags=re.M)) - {"constructor"}
        r = _node("const m=require(%r);console.log(JSON.stringify(Object.getOwnPropertyNames("
                  "m.%s.prototype)))" % (ADDON, name))
        assert r.returncode == 0, r.stderr
        have = set(json.loads(r.stdout)) - {"constructor"}
        assert declared == have, (name, declared ^ have)


def test_method_list_is_the_reference_ffi(oracle):
    """WASM_METHODS above is checked against the reference source when it is mounted."""
    src = "/root/reference/physics-engine/gravitas-wasm/src/lib.rs"
    if not os.path.exists(src):
        pytest.skip("reference not mounted")
    names = re.findall(r"pub fn (\w+)\(", open(src).read())
    want = set(names) - {"init_hooks", "new"}
    assert want <= set(WASM_METHODS), want - set(WASM_METHODS)


def test_no_device_is_an_exception_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = _node("const m=require(%r);try{new m.PhysicsEngine(1,0.9);console.log('created')}"
              "catch(e){console.log('threw:'+e.message)}" % ADDON)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("threw:") and "no HIP device" in r.stdout


@pytest.mark.gpu
def test_smoke_js_matches_oracle(oracle):
    r = subprocess.run([NODE, os.path.join(ROOT, "napi", "smoke.js")], capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr + r.stdout
    res = json.loads(r.stdout.strip().splitlines()[-1])
    L = oracle.lib()
    m = oracle.metric(oracle.KERR_BL, 1.0, 0.9)
    assert abs(res["horizon"] - L.orc_event_horizon(m)) < 1e-12
    assert abs(res["isco"] - L.orc_isco(m, 0)) < 1e-12
    assert res["layout"] == [0, 64, 128, 256, 2048]
    assert abs(res["sab_horizon"] - res["horizon"]) < 1e-6 and abs(res["sab_isco"] - res["isco"]) < 1e-6
    assert res["sab_points"] >= 32
    ref = oracle.integrate_ray_relativistic(1.0, 0.9, [0, 20, math.pi / 2, 0, -1, -1, 0, 3.5], 10000,
                                            1e-8, True)
    assert np.allclose(res["ray"], ref, rtol=1e-6, atol=1e-6)
    assert res["ray_again"] == res["ray"] and res["ray_fast"] != res["ray"]      # FAST and back to STRICT
    assert np.allclose(res["ray_fast"], res["ray"], rtol=1e-6, atol=1e-6) and res["bad_arith"] == "TypeError"
    assert res["echo"] == [1, 2, 3]                                  # lib.rs:429-431
    lut = oracle.blackbody_lut(8, 2, 1e5).reshape(-1)
    assert np.allclose(res["lut0"], lut[28:32], rtol=1e-5)
    assert res["disk_lut_len"] == 512 and res["disk_lut_max"] == 1.0
    assert res["shadow_pts"] in (32, 64)
    sh = oracle.bardeen_shadow(1.0, 0.9, math.pi / 2, 32)[:, 0]
    assert np.allclose(res["shadow_shift"], [sh.min(), sh.max()], atol=1e-5)
    assert res["embedding_len"] == 3 * 24 * 16 and res["ergosphere_len"] == 3 * 17 * 12
    assert abs(res["kretschner"] - L.orc_kretschner_kerr(6.0, math.pi / 2, 1.0, 0.9)) < 1e-12
    assert abs(res["tilt"] - L.orc_light_cone_tilt_bl(6.0, math.pi / 2, 1.0, 0.9)) < 1e-12
    assert abs(res["omega"] - L.orc_frame_dragging_omega(6.0, math.pi / 2, 1.0, 0.9)) < 1e-12
    assert abs(res["flamm"] - L.orc_flamm_height(100.0, 1.0)) < 1e-12
    assert abs(res["proper"] - L.orc_proper_distance(4.0, 20.0, 500, 1.0, 0.9)) < 1e-10
    assert res["curvature_len"] == res["tilt_len"] == res["drag_len"] == 3 * 8 * 5
    assert res["webgl"]["len"] == 64 * 36 * 4 and res["webgl"]["alpha"] == 1.0 and 0.0 < res["webgl"]["max"] <= 1.0
    # renderFrame: same frame through the oracle (96x54, a = 0.9, camera of smoke.js)
    cam = oracle.camera_look_at((59.55, -7.31, 0.0), aspect=96 / 54)
    fr = oracle.render_frame(cam, oracle.frame_params(96, 54, spin=0.9), None, nthreads=4)
    assert res["frame"]["rays"] == 96 * 54 and res["frame"]["alpha0"] == 1.0
    assert res["frame"]["acceptedSteps"] == int(fr["steps"].sum())
    assert res["frame"]["lit"] == int((fr["rgba"].reshape(-1, 4)[:, :3].sum(axis=1) > 0).sum())
    # arena slots (1 MiB / 10 KiB = 102): freed engines give their slot back, live engines never share
    # one, and running out is an exception instead of an alias of slot 0
    ar = res["arena"]
    assert ar["reusedSlots"] == 1 and ar["reuseAvoidsLive"]
    assert ar["liveEngines"] == 101 and ar["distinctLivePtrs"] == 101 and ar["liveAvoidFirst"]
    assert ar["exhausted"] and "arena exhausted" in ar["exhausted"]


@pytest.mark.gpu
def test_worker_and_bridge_sab_protocol(oracle):
    """napi/worker_protocol.js speaks physics.worker.ts's writer and physics-bridge.ts's seqlock
    reader against the addon; every published tick must equal the oracle's tick_sab block."""
    r = subprocess.run([NODE, os.path.join(ROOT, "napi", "worker_protocol.js")], capture_output=True,
                       text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr + r.stdout
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["torn"] == 0 and [t["seq"] for t in res["ticks"]] == [2, 4, 6, 8]
    at = res["attach"]  # attach_sab (lib.rs:74): tick_sab moves, get_sab_ptr does not; bad offsets throw RangeError
    assert at["ptr_unchanged"] and at["published_there"] and at["control_consumed_there"] and at["own_block_untouched"]
    assert at["errors"] == [True] * 6
    o = oracle.sab_engine(1.0, 0.9)
    o.camera.auto_spin = 1
    o.camera.position[0], o.camera.position[1], o.camera.position[2] = 3.0, 4.0, 12.0
    for k, t in enumerate(res["ticks"]):
        o.sab[1], o.sab[3] = 0.5 * k, -0.1
        want = oracle.tick_sab(o, 0.016)
        assert t["finite"] and t["inputs_consumed"]
        assert np.allclose(t["camera"], want[64:76], rtol=1e-6, atol=1e-6)
        assert np.allclose(t["physics"], want[128:256], rtol=1e-6, atol=1e-6)


def _bulk_rays(n):
    """The initial states napi/bulk.js builds (same IEEE operations)."""
    s = np.zeros((n, 8))
    for i in range(n):
        u = (i + 0.5) / n
        s[i] = [0, 20 + 40 * ((i * 7919) % n) / n, math.pi / 2 - 0.4 + 0.8 * u, 0.1 * i, -1, -1, 0.3 - 0.6 * u,
                -8 + 16 * u]
    return s


@pytest.mark.gpu
def test_bulk_js_batch_async_and_multi_device_entries(oracle, tmp_path):
    """napi/bulk.js: 4096 geodesics through ONE integrate_batch call equal the oracle's integrate()
    bit for bit (STRICT is the default contract of the entry, as for integrate_ray_relativistic); the
    *Async forms return the same bits while the caller's timer keeps ticking; renderFrame over
    virtual ranks / into caller-owned pinned memory equals the one-device frame bit for bit."""
    out = tmp_path / "bulk.json"
    r = subprocess.run([NODE, os.path.join(ROOT, "napi", "bulk.js"), str(out)], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr + r.stdout[-2000:]
    res = json.loads(out.read_text())
    n = res["batch"]["n"]
    init = _bulk_rays(n)
    assert np.array_equal(np.array(res["batch"]["init"]), init.reshape(-1)[:64])
    ref = oracle.integrate_batch(oracle.metric(oracle.KERR_KS, 1.0, 0.9),
                                 oracle.options(max_steps=2000, tolerance=1e-8), init, nthreads=8)
    got = np.array(res["batch"]["states"]).reshape(n, 8)
    assert np.array_equal(got.view(np.uint64), ref["states"].view(np.uint64))
    assert np.array_equal(np.array(res["batch"]["steps"], np.uint32), ref["steps"])
    assert np.array_equal(np.array(res["batch"]["term"], np.uint8), ref["term"])
    assert np.array_equal(np.array(res["batch"]["drift"]).view(np.uint64), ref["drift"].view(np.uint64))
    assert len(set(res["batch"]["term"])) >= 2  # captured and escaped rays both present
    # the one-ray FFI entry agrees with the batch on the same rays (same kernel body)
    assert np.array_equal(np.array(res["single"]), got[:32])
    a = res["async"]
    assert a["batch_equal"] and a["frame_rays"] == 640 * 360 and a["frame_steps"] > 0
    assert a["ticks_during"] >= 1  # the event loop ran while the pool thread waited for the GPU
    f = res["frames"]
    assert f["ranks_equal"] and f["async_ranks_equal"] and f["pinned_equal"] and f["pinned_is_out"]
    assert f["steps1"] == f["steps4"] and f["devices4"] == 4 and f["update_reaches_ranks"]
    # renderFrame({exchange: "rgba16f"}): the gather in the compute pass's rgba16float format (renderer.ts:163-176)
    fh = res["frames_half"]
    assert fh["equal_rounded"] and fh["differs_from_f32"] and fh["back_to_f32"], fh
    am = res["async_memory"]  # the async forms never hold pointers into memory JS can reach meanwhile
    assert am["out_filled"] and am["input_copy"] and am["detached_rejected"] and "detached" in am["detached_rejected"]
    e = res["errors"]
    assert all(e["sync"]) and "8 n" in e["sync"][0] and "out of range" in e["sync"][1] and "out must be" in e["sync"][2]
    assert e["async_rejected"] and res["empty"] == 0 and res["free_while_pending"] == 100
    # per-ray cost from JS: a 4096-ray call must beat the one-ray entry by two orders of magnitude
    assert res["us_per_ray_batch"] * 100 < res["single_ms_per_ray"] * 1e3
    # recordPath: Trajectory.path from JS = the oracle's orc_integrate_path rows, bit for bit (spin 0.5 by then)
    pth = res["paths"]
    mp = pth["maxPoints"]
    rows = np.array(pth["rows"]).reshape(16, mp, 8)
    m5 = oracle.metric(oracle.KERR_KS, 1.0, 0.5)
    for i in range(16):
        t, ref_path = oracle.integrate_path(np.array(pth["init"][8 * i:8 * i + 8]), m5,
                                            oracle.options(max_steps=2000, tolerance=1e-8), cap=2001)
        assert pth["counts"][i] == ref_path.shape[0] == pth["steps"][i] + 1
        k = min(mp, ref_path.shape[0])
        assert np.array_equal(rows[i, :k].view(np.uint64), ref_path[:k].view(np.uint64)), i
