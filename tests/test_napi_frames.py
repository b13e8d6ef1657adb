"""BASELINE's frames driven from the host north_star names -- JavaScript through the N-API addon --
must be the frames bench.py times through Python ctypes: same C ABI underneath, so the pixel bytes
(SHA-256) and the accepted-step totals are required to be EQUAL at full size, for every form the
addon offers: device-resident (renderFrame({keepOnDevice}) -> DeviceImage), asynchronous into
page-locked memory, and the synchronous host form.  (reference callers: src/rendering/webgpu/renderer.ts:280-411,
src/workers/physics.worker.ts:111-176, gravitas-wasm/src/lib.rs:422-464)"""
import hashlib
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADDON = os.path.join(ROOT, "napi", "blackhole_physics.node")
NODE = shutil.which("node")

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(NODE is None or not os.path.exists(ADDON), reason="node or the built addon is not available")]


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def js(tmp_path_factory):
    out = tmp_path_factory.mktemp("napi") / "frames.json"
    r = subprocess.run([NODE, os.path.join(ROOT, "napi", "frames_check.js"), str(out)], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr + r.stdout[-2000:]
    return json.loads(out.read_text())


def test_4k_frame_from_node_is_the_ctypes_frame_bit_for_bit(engine_mod, js):
    import torch
    bh = engine_mod
    W, H = js["W"], js["H"]
    assert (W, H) == (3840, 2160)
    th = np.deg2rad(97.0)
    eye = (60.0 * np.sin(th), 60.0 * np.cos(th), 0.0)
    with bh.PhysicsEngine(1.0, 0.999) as eng:
        rgba = torch.zeros(W * H, 4, dtype=torch.float32, device="cuda")
        eng.render_frame_device(bh.camera_look_at(eye, aspect=W / H),
                                bh.render_params(W, H, arith=bh.ARITH_FAST, tolerance=1e-8), rgba)
        st = eng.frame_stats()
        want = _sha(rgba.cpu().numpy())
    d = js["device"]
    assert d["queued"] and not d["has_rgba"] and (d["w"], d["h"], d["bytes"]) == (W, H, W * H * 16)
    assert d["steps"] == st.accepted_steps and d["rays"] == W * H
    assert d["sha"] == want and d["sha_again"] == want
    assert d["read_async"]["sha"] == want and d["read_async"]["same_array"]
    a = js["async"]
    assert a["sha0"] == want and a["sha1"] == want and a["is_out"]
    assert a["steps0"] == a["steps1"] == st.accepted_steps
    assert a["plain_sha"] == want and a["plain_steps"] == st.accepted_steps
    assert js["host"]["sha"] == want and js["host"]["steps"] == st.accepted_steps


def test_1080p_marches_and_post_chain_from_node(engine_mod, js):
    import torch
    bh = engine_mod
    W, H = 1920, 1080
    th = np.deg2rad(97.0)
    eye = (60.0 * np.sin(th), 60.0 * np.cos(th), 0.0)
    with bh.PhysicsEngine(1.0, 0.999) as eng:
        rgba = torch.zeros(H, W, 4, dtype=torch.float32, device="cuda")
        gp = bh.glsl_params(W, H, 1.0, 0.999, max_ray_steps=512, arith=bh.ARITH_FAST)
        tot = eng.render_frame_glsl(gp, rgba)
        assert js["glsl"]["steps"] == tot == js["glsl"]["host_steps"]
        assert js["glsl"]["sha"] == _sha(rgba.cpu().numpy()) == js["glsl"]["host_sha"]
        bloom = torch.zeros_like(rgba)
        eng.post_bloom(W, H, rgba, bloom, arith=bh.ARITH_FAST)
        torch.cuda.synchronize()
        assert js["bloom"]["sha"] == _sha(bloom.cpu().numpy())
        rgba2 = torch.zeros_like(rgba)
        eng.render_frame_glsl(bh.glsl_params(W, H, 1.0, 0.999, max_ray_steps=512, arith=bh.ARITH_FAST, time=0.5), rgba2)
        taa = torch.zeros_like(rgba)
        eng.post_taa_resolve(W, H, rgba2, rgba, taa, arith=bh.ARITH_FAST)
        torch.cuda.synchronize()
        assert js["taa"]["sha"] == _sha(taa.cpu().numpy())
        wp = bh.wgsl_params(W, H, bh.camera_look_at(eye, aspect=W / H), 1.0, 0.999, max_steps=512,
                            arith=bh.ARITH_FAST_PACKED)
        tot = eng.render_frame_wgsl(wp, rgba)
        assert js["wgsl"]["steps"] == tot and js["wgsl"]["sha"] == _sha(rgba.cpu().numpy())
    with bh.PhysicsEngine(1.0, 0.999) as eng:   # the WebGL renderer: frames 2 and 3 of a 4-frame sequence
        scr = torch.zeros(360, 640, 4, dtype=torch.float32, device="cuda")
        shas = []
        for i in range(4):
            gp = bh.glsl_params(640, 360, 1.0, 0.9, arith=bh.ARITH_FAST, time=0.1 * i)
            eng.webgl_render(gp, scr, bloom=True, camera_moving=False)
            shas.append(_sha(scr.cpu().numpy()))
        assert js["webgl"]["sha_frame2"] == shas[2] and js["webgl"]["sha_frame3"] == shas[3]


def test_image_errors_and_lifetime_from_node(js):
    e = js["errors"]
    assert all(e["sync"]), e
    assert "into an image" in e["sync"][0] and "DeviceImage" in e["sync"][1]
    assert "allocPinned" in e["sync"][2] and "aliases" in e["sync"][3]
    assert e["async_keep"] and "synchronous" in e["async_keep"]
    lt = js["lifetime"]
    assert lt["read_after_engine_free"] == 64 * 36 * 4 and "no device memory" in lt["freed_msg"]
