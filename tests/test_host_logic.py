"""CPU tests of the host side: the C-ABI library loads without a GPU, exports
every symbol the header declares, fails loudly without a device, and the tile
partition / camera / defaults agree with the oracle's statements."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "gravitas_abi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(grv_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(engine_mod):
    lib = engine_mod.load_library()
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libgravitas_hip.so does not export %s" % n
    assert lib.grv_abi_version() == 8


def test_no_device_fails_loudly(engine_mod):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(engine_mod.GravitasError):
        engine_mod.PhysicsEngine(1.0, 0.9)


_RCCL_PROBE = r"""
import sys
sys.path.insert(0, %(root)r)
import ctypes as C
import blackhole_simulation_amd as bh
L = bh.load_library()
for attempt in range(3):   # a failed bind must stay a clean failure: never "already loaded" over null entry points
    v, why = bh.rccl_probe()
    print("PROBE", v, why)
h = C.c_void_p()
rc = L.grv_engine_create_multi(1.0, 0.5, 0b11, bh.TRANSPORT_RCCL, C.byref(h))
print("CREATE", rc, bool(h.value), L.grv_multi_create_error().decode())
"""


@pytest.mark.parametrize("lib,expect", [("/nonexistent/librccl.so.1", "dlopen(/nonexistent/librccl.so.1)"),
                                        ("libm.so.6", "librccl lacks ncclCommInitAll")])
def test_rccl_transport_without_a_usable_librccl_is_a_clean_error(engine_mod, lib, expect):
    """A host whose librccl is absent (or is not RCCL) gets status codes and the loader's text, not a
    crash: dlerror() is read once, and a library that lacks an entry point is closed again instead of
    staying half-bound (csrc/engine_multi.hip RcclApi::load)."""
    import subprocess
    import sys
    env = dict(os.environ, GRV_RCCL_LIBRARY=lib)
    r = subprocess.run([sys.executable, "-c", _RCCL_PROBE % {"root": ROOT}], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    probes = [ln for ln in r.stdout.splitlines() if ln.startswith("PROBE")]
    assert len(probes) == 3 and all(ln.startswith("PROBE 0 ") and expect in ln for ln in probes), r.stdout
    create = [ln for ln in r.stdout.splitlines() if ln.startswith("CREATE")][0].split(" ", 3)
    # no GPU here: refused for the missing device before RCCL is looked at; on a GPU box: for RCCL
    assert create[1] == "2" and create[2] == "False" and create[3], r.stdout


def test_rccl_probe_reports_the_bound_version(engine_mod):
    v, why = engine_mod.rccl_probe()
    if v == 0:
        pytest.skip("no librccl on this host: %s" % why)
    assert v >= 20000 and why == ""  # ncclGetVersion's code: major * 10000 + minor * 100 + patch


def test_package_has_no_oracle_dependency():
    """The product must never route through oracle/ (or any CPU fallback)."""
    for top in ("blackhole-simulation_amd", "napi", "tools", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".c", ".js", ".sh")) or f == "Makefile":
                    src = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "pyoracle" not in src and "gravitas_oracle" not in src and "orc_" not in src, f
    # bench.py touches the oracle in its cpu_baseline leg only
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def cpu_baseline_c3("):src.index("def committed_pmc(")]
    assert "pyoracle" in body and "pyoracle" not in src.replace(body, "")


def test_sab_layout(engine_mod):  # gravitas-wasm/src/lib.rs:36-40, 411-419
    a = (C.c_size_t * 5)()
    engine_mod.load_library().grv_get_sab_layout(a)
    assert list(a) == [0, 64, 128, 256, 2048]


def test_option_defaults_match_reference(engine_mod, oracle):  # integrator.rs:35-47
    o = engine_mod.engine.default_options()
    ro = oracle.lib().orc_options_default()
    assert (o.method, o.tolerance, o.initial_step, o.max_steps, o.escape_radius,
            o.renormalize_interval) == (ro.method, ro.tolerance, ro.initial_step, ro.max_steps,
                                        ro.escape_radius, ro.renormalize_interval)
    assert o.metric_kind == engine_mod.KERR_KS and o.arith == engine_mod.ARITH_STRICT


def test_camera_helper_matches_oracle_statement(engine_mod, oracle):
    eye = (60 * np.sin(1.7), 60 * np.cos(1.7), 3.0)
    a = engine_mod.camera_look_at(eye, aspect=1.5)
    b = oracle.camera_look_at(eye, aspect=1.5)
    for f in ("position", "inv_view", "inv_proj", "pixel_offset"):
        assert list(getattr(a, f)) == list(getattr(b, f)), f


def test_camera_from_webgpu_uniform_block(engine_mod):  # src/types/webgpu.ts:95-116
    u = np.arange(88, dtype=np.float32)
    cam = engine_mod.Camera()
    engine_mod.load_library().grv_camera_from_uniforms(u.ctypes.data_as(C.c_void_p), C.byref(cam))
    assert list(cam.inv_view) == list(range(32, 48)) and list(cam.inv_proj) == list(range(48, 64))
    assert list(cam.position) == [80.0, 81.0, 82.0]


@pytest.mark.parametrize("w,h,world", [(3840, 2160, 1), (3840, 2160, 8), (100, 70, 3), (64, 64, 2)])
def test_tile_partition_covers_every_pixel_once(engine_mod, w, h, world):
    from blackhole_simulation_amd import distributed as D
    p = engine_mod.render_params(w, h)
    counts = np.zeros((h, w), np.int32)
    for r in range(world):
        rp = D.rank_params(p, world, r)
        n = engine_mod.load_library().grv_frame_ray_count(C.byref(rp))
        tiles = D.tiles_of_rank(w, h, world, r)
        assert n == (w * h if world == 1 else len(tiles) * 4096)
        if world == 1:
            counts += 1
            continue
        packed = np.ones((n, 1), np.int32)
        counts += engine_mod.unpack_tiles(rp, r, packed, 1, np.int32)[..., 0]
    assert np.all(counts == 1)
    assert sum(len(D.tiles_of_rank(w, h, world, r)) for r in range(world)) == D.tiles_total(w, h, world)
    assert engine_mod.load_library().grv_tile_pitch(w, world) == D.tile_pitch(w, world)


@pytest.mark.parametrize("w,h,world", [(3840, 2160, 8), (7680, 4320, 4), (1920, 1080, 2), (200, 130, 3),
                                       (64, 64, 2), (100, 70, 1), (640, 360, 5), (7680, 4320, 6)])
def test_tile_deal_exports_match_their_definition(engine_mod, w, h, world):
    """The library holds the ONE implementation of the tile deal (grv_tile_pitch, grv_tiles_total,
    grv_tiles_of_rank, grv_tile_origin, grv_max_tiles_per_rank); blackhole_simulation_amd.distributed
    is thin calls into it.  The definition, restated here as the pin (row-major 64x64 grid of
    physics-engine/_legacy_src/tiling.rs:38-56 on a pitch coprime with the rank count, id k -> rank
    k mod world):"""
    import inspect
    import math
    from blackhole_simulation_amd import distributed as D
    pitch = (w + 63) // 64
    while world > 1 and math.gcd(pitch, world) != 1:
        pitch += 1
    total = pitch * ((h + 63) // 64)
    lib = engine_mod.load_library()
    assert lib.grv_tile_pitch(w, world) == pitch == D.tile_pitch(w, world)
    assert lib.grv_tiles_total(w, h, world) == total == D.tiles_total(w, h, world)
    assert lib.grv_max_tiles_per_rank(w, h, world) == (total + world - 1) // world == D.max_tiles_per_rank(w, h, world)
    seen = []
    for r in range(world):
        ids = D.tiles_of_rank(w, h, world, r)
        assert ids == list(range(r, total, world))
        assert lib.grv_tiles_of_rank(w, h, world, r, None, 0) == len(ids)  # size query writes nothing
        short = (C.c_uint32 * 2)(0xFFFFFFFF, 0xFFFFFFFF)
        assert lib.grv_tiles_of_rank(w, h, world, r, short, 1) == len(ids) and short[1] == 0xFFFFFFFF  # capacity kept
        p = engine_mod.render_params(w, h)
        rp = D.rank_params(p, world, r)
        assert lib.grv_frame_ray_count(C.byref(rp)) == (w * h if world == 1 else len(ids) * 4096)
        seen += ids
    assert sorted(seen) == list(range(total))
    assert lib.grv_tiles_of_rank(w, h, world, world, None, 0) == 0  # no such rank
    for t in (0, 1, pitch - 1, pitch, total - 1):
        assert D.tile_origin(t, w, world) == ((t % pitch) * 64, (t // pitch) * 64)
    # thin: the module's deal functions hold no arithmetic of their own
    for fn in (D.tile_pitch, D.tiles_total, D.tiles_of_rank, D.tile_origin, D.max_tiles_per_rank):
        src = inspect.getsource(fn)
        assert "load_library()" in src and "gcd" not in src and "//" not in src and "%" not in src.split('"""')[-1], fn


def test_verification_hooks_are_locked_until_the_key_is_given():
    """grv_test_set_try_bound / grv_multi_test_self_exchange must not answer a stray call: a freshly
    loaded library refuses them, a wrong key does not unlock, the right one does (own process: the
    lock is per process)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import blackhole_simulation_amd as bh\n"
            "L = bh.load_library()\n"
            "print('LOCKED', L.grv_test_hooks_unlocked(), L.grv_test_hooks_unlock(1234), L.grv_test_hooks_unlocked())\n"
            "print('NULL', L.grv_test_set_try_bound(None, 5), L.grv_multi_test_self_exchange(None, 1), L.grv_test_try_bound(None))\n"
            "bh.unlock_test_hooks()\n"
            "print('OPEN', L.grv_test_hooks_unlocked())\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "LOCKED 0 1 0" in r.stdout and "NULL 1 1 0" in r.stdout and "OPEN 1" in r.stdout, r.stdout


def test_unpack_places_pixels_row_major(engine_mod):
    from blackhole_simulation_amd import distributed as D
    w, h, world = 200, 130, 3
    p = engine_mod.render_params(w, h)
    truth = np.arange(w * h, dtype=np.int32).reshape(h, w)
    img = np.zeros((h, w, 1), np.int32)
    for r in range(world):
        rp = D.rank_params(p, world, r)
        tiles = D.tiles_of_rank(w, h, world, r)
        packed = np.zeros((len(tiles) * 4096, 1), np.int32)
        for tl, t in enumerate(tiles):
            x0, y0 = D.tile_origin(t, w, world)
            blk = np.zeros((64, 64), np.int32)
            sub = truth[y0:y0 + 64, x0:x0 + 64]
            blk[:sub.shape[0], :sub.shape[1]] = sub
            packed[tl * 4096:(tl + 1) * 4096, 0] = blk.reshape(-1)
        img += engine_mod.unpack_tiles(rp, r, packed, 1, np.int32)
    assert np.array_equal(img[..., 0], truth)


@pytest.mark.parametrize("w,world", [(3840, 2), (3840, 4), (3840, 8), (7680, 8), (7680, 4), (1920, 8),
                                     (200, 3), (640, 5), (64, 2), (3840, 1), (7680, 6)])
def test_tile_deal_shifts_from_row_to_row(w, world):
    """The pitch is coprime with the rank count, so no rank owns whole tile columns: every rank
    meets every tile column within `world` consecutive rows, and the pad stays small."""
    import math
    from blackhole_simulation_amd import distributed as D
    p = D.tile_pitch(w, world)
    across = (w + 63) // 64
    assert p >= across and math.gcd(p, world) == 1 and p - across < max(world, 2)
    if world == 1:
        assert p == across
        return
    for col in range(min(across, 9)):
        owners = {(row * p + col) % world for row in range(world)}
        assert owners == set(range(world)), (col, owners)


def test_bench_helpers():
    """bench.py pieces that run without a GPU: the weak-scaling grid, the host-core count that
    honours the cgroup quota, the committed PMC traffic record bench.py quotes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert all(gx * gy == n for n, (gx, gy) in b.GRID.items()) and set(b.GRID) == {1, 2, 4, 8}
    assert 1 <= b.usable_cores() <= (os.cpu_count() or 1)
    assert (b.B_STEP["c3"], b.B_RAY["c3"], b.HBM_PEAK_GBS) == (144, 96, 8000.0)  # SURVEY 8(d), microarch guide
    assert b.B_STEP["c4"] == 72 and b.FP64_PEAK_TFLOPS == 78.6
    assert b.cpu_model()
    import blackhole_simulation_amd as bh
    if not os.path.exists(bh.library_path()):
        bh.build_library()
    # the committed PMC figures are quoted only for the code object they were measured on
    t, src = b.committed_pmc("integrate_segment_kernel<1,1,0>", bh.library_path())
    if t is None:
        assert src.startswith("stale"), src  # the kernel changed since the committed pass: dropped
    else:
        assert t["hbm_bytes_per_launch"] > 1e9 and 0.5 < t["valu"]["issue_frac"] <= 1.0
        # measured HBM bytes against the layout's 92 B read + 76 B written per slot: within 2 % when the waves
        # start in slot order (rounds 2-5); with the longest-first order of round 6 neighbouring waves no longer
        # run side by side and L2 merges fewer of the partial-line stores (crossing records, 4-byte arrays):
        # +13 % written, +7 % fetched -- 1.54 GB per 27 ms launch, 0.7 % of the HBM peak either way
        assert -0.02 < t["hbm_bytes_per_launch"] / t["expected_from_layout_bytes"] - 1.0 < 0.15
    none, why = b.committed_pmc("no_such_kernel", bh.library_path())
    assert none is None and "no committed" in why


def test_roofline_block_is_always_a_number_against_the_bound_that_binds():
    """bench.py's roofline object: bound = the vector ALU of the dtype, frac =
    flops / launch time / vector peak and never null -- counted flops when a counter pass of the code
    object on disk is committed, SURVEY 8(d)'s algorithmic flops otherwise; the 8(d) byte figure only
    as the labelled hbm_nominal sub-block."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod3", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    steps, rays, ms = 1511626085.0, 8294400, 27.2814
    pmc = {"pass_id": "rXX-test", "frame": [3840, 2160], "hbm_bytes_per_launch": 1410739552,
           "valu": {"issue_frac": 0.9488, "flops_counted_per_launch": 1286602997696.0}}
    r = b.roofline_block("c3", False, "integrate_segment_kernel<1,1,0>", ms, 1.0, steps, rays, pmc,
                         "profiles/traffic.json (code object x)", True, 0, "events")
    assert r["bound"] == "fp64_valu" and r["unit"] == "TFLOP/s" and r["peak"] == 78.6
    assert r["flops_source"] == "counted" and abs(r["achieved"] - 1286602997696.0 / (ms * 1e-3) / 1e12) < 0.01
    assert abs(r["frac"] - r["achieved"] / 78.6) < 1e-3 and 0.55 < r["frac"] < 0.65
    assert r["traffic"] == 1410739552 and r["valu_issue_frac"] == 0.9488 and r["traffic_pass"] == "rXX-test"
    h = r["hbm_nominal"]
    assert h["algorithmic_bytes_per_launch"] == 218470418640 and abs(h["nominal_frac"] - 1.001) < 2e-3
    assert 45 < h["hbm_measured_GBps"] < 60 and "registers" in h["reason"]
    # no committed pass for the library on disk: SURVEY 8(d)'s 1.3 kflop per f64 step, still a number
    r = b.roofline_block("c3", False, "k", ms, 1.0, steps, rays, None, "stale: ...", False, 0, "events")
    assert r["flops_source"] == "algorithmic" and r["flops_per_ray_step"] == 1300.0
    assert isinstance(r["frac"], float) and 0 < r["frac"] <= 1.0 and r["traffic"] is None and r["valu_issue_frac"] is None
    # the K-try schedule: per-launch figures, its own FETCH / WRITE pass when one is committed
    pmc["segment_tries_16"] = {"hbm_bytes_per_launch": 700000000, "launches_per_frame": 32}
    r = b.roofline_block("c3", False, "k", 1.0, 32.0, steps, rays, pmc, "profiles/traffic.json", True, 16, "events")
    assert r["traffic"] == 700000000 and r["ray_steps_per_launch"] == int(steps / 32) and r["valu_issue_frac"] is None
    assert "segment" in r["traffic_source"]
    # f32 marches: the FP32 vector peak
    for cfg, glsl, per in (("c4", False, 500.0), ("c2", False, 500.0), ("c2", True, 150.0)):
        r = b.roofline_block(cfg, glsl, "k", 2.0, 1.0, 8e8, 2073600, None, "none", False, 0, "events")
        assert r["bound"] == "fp32_valu" and r["peak"] == 157.3 and r["flops_per_ray_step"] == per
        assert isinstance(r["frac"], float) and r["hbm_nominal"]["nominal_frac"] > 0


def test_stale_pmc_figures_are_dropped(tmp_path, monkeypatch):
    """A traffic.json stamped with another code hash must not be quoted."""
    import importlib.util
    import json
    import shutil
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    import blackhole_simulation_amd as bh
    fake = tmp_path / "repo"
    (fake / "profiles").mkdir(parents=True)
    shutil.copytree(os.path.join(ROOT, "tools"), fake / "tools")
    json.dump({"format": 2, "kernels": {"integrate_segment_kernel<1,1,0>": {
        "code_hash": "0123456789abcdef", "hbm_bytes_per_launch": 1}}}, open(fake / "profiles" / "traffic.json", "w"))
    monkeypatch.setattr(b, "ROOT", str(fake))
    t, why = b.committed_pmc("integrate_segment_kernel<1,1,0>", bh.library_path())
    assert t is None and why.startswith("stale")


def test_kernel_code_hash_tracks_the_code_object(engine_mod):
    sys_path = os.path.join(ROOT, "tools")
    import sys
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    import kernel_resources as kr
    lib = engine_mod.library_path()
    a = kr.kernel_code_hash(lib, "integrate_segment_kernel<1,1,0>")
    b2 = kr.kernel_code_hash(lib, "integrate_segment_kernel<1,0,0>")
    assert a and b2 and a != b2 and len(a) == 16
    assert kr.kernel_code_hash(lib, "integrate_segment_kernel<1,1,0>") == a  # deterministic
    assert kr.kernel_code_hash(lib, "no_such_kernel") is None


def test_shadow_curve_capacity_is_honoured(engine_mod):
    """grv_compute_shadow_curve never writes past out_capacity (on-axis observers double the curve)."""
    lib = engine_mod.load_library()
    # no engine without a device: the size query must tolerate a NULL handle (returns 0)
    assert lib.grv_compute_shadow_curve(None, 0.0, 16, None, 0) == 0


def test_bench_gpus_n_is_never_silently_one_gpu():
    """`python bench.py --gpus N` without a launcher starts its own N ranks; with fewer devices
    than N it must exit non-zero and print no result line (a bare --gpus 8 used to run
    one GPU and print n_gpus 1)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GRV_BENCH_ONE_DEVICE", "GRV_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert "HIP device(s) visible" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_refuses_what_it_cannot_run_before_touching_a_device():
    """`python bench.py --gpus N` on a host with fewer devices exits 2 with the device count on stderr
    and no JSON line (never a line for fewer GPUs than asked); --config c2 is a one-GPU configuration."""
    import subprocess
    import sys
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GRV_BENCH_ONE_DEVICE")}
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", str(have + 1)], capture_output=True, text=True, timeout=300,
                       env=env, cwd=ROOT)
    assert r.returncode == 2 and "only %d HIP device" % have in r.stderr, (r.returncode, r.stderr[-500:])
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    r = subprocess.run([sys.executable, bench, "--config", "c2", "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "one GPU" in r.stderr
    r = subprocess.run([sys.executable, bench, "--config", "c3", "--arith", "packed"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "packed" in r.stderr


def test_committed_counter_passes_are_of_the_library_in_the_tree(engine_mod):
    """profiles/traffic.json is stamped with the code hash of every kernel its PMC passes ran on; the
    roofline blocks bench.py prints quote it only for the same kernel.  The stamps must be those of the
    library in the tree, or a bench line would say `stale` -- so a kernel change without new passes
    fails here, on the CPU, before it reaches a GPU box."""
    import json
    import sys
    sys_path = os.path.join(ROOT, "tools")
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    import kernel_resources as kr
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    lib = engine_mod.library_path()
    assert len(t["kernels"]) >= 6
    for name, ent in t["kernels"].items():
        assert kr.kernel_code_hash(lib, name.split("@")[0]) == ent["code_hash"], name
