"""Static checks on the gfx950 code objects inside libgravitas_hip.so (no GPU needed): the AMDGPU
metadata note of every kernel is decoded (clang offload bundle -> ELF -> NT_AMDGPU_METADATA msgpack)
and the properties the performance numbers rest on are asserted, so that a change which silently
costs the headline kernel a wave of occupancy or spills a hot kernel to scratch fails here."""
import os
import struct

import pytest

msgpack = pytest.importorskip("msgpack")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _kernels(path):
    data = open(path, "rb").read()
    out, pos = {}, 0
    while True:
        base = data.find(MAGIC, pos)
        if base < 0:
            break
        pos = base + 1
        (n,) = struct.unpack_from("<Q", data, base + 24)
        off = base + 32
        for _ in range(n):
            o, s, t = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + t].decode()
            off += t
            if "amdgcn" not in triple or s == 0:
                continue
            assert "gfx950" in triple, triple          # this library is built for one target
            elf = data[base + o:base + o + s]
            assert elf[:4] == b"\x7fELF"
            (shoff,) = struct.unpack_from("<Q", elf, 0x28)
            shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
            for k in range(shnum):
                sh = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + k * shentsize)
                if sh[1] != 7:      # SHT_NOTE
                    continue
                p, end = sh[4], sh[4] + sh[5]
                while p < end:
                    namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
                    p += 12
                    name = elf[p:p + namesz]
                    p += (namesz + 3) & ~3
                    desc = elf[p:p + descsz]
                    p += (descsz + 3) & ~3
                    if name.startswith(b"AMDGPU") and ntype == 32:
                        md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                        for kd in md["amdhsa.kernels"]:
                            out[kd[".name"]] = kd
    return out


@pytest.fixture(scope="module")
def kernels(engine_mod):
    path = engine_mod.library_path()
    assert os.path.exists(path)
    ks = _kernels(path)
    assert len(ks) >= 40
    return ks


def _find(kernels, *parts):
    hits = [kd for n, kd in kernels.items() if all(p in n for p in parts)]
    assert hits, parts
    return hits


def test_headline_kernel_keeps_three_waves_per_simd(kernels):
    # integrate_segment_kernel<KerrSchild, FAST, RKF45>: 512 VGPRs per SIMD lane / 3 waves -> <= 168,
    # no AGPRs, no scratch (2 waves/SIMD costs ~2.5 %)
    (kd,) = _find(kernels, "integrate_segment_kernelILi1ELi1ELi0E")
    assert kd[".vgpr_count"] + kd.get(".agpr_count", 0) <= 168, kd[".vgpr_count"]
    assert kd[".private_segment_fixed_size"] == 0 and kd.get(".vgpr_spill_count", 0) == 0
    assert kd[".wavefront_size"] == 64 and kd[".max_flat_workgroup_size"] == 256


def test_no_hot_kernel_uses_scratch(kernels):
    for part in ("integrate_segment_kernel", "integrate_refill_kernel", "wgsl_symplectic_pk_kernel",
                 "init_from_pixels_kernel",
                 "init_from_states_kernel", "finalize_frame_kernel", "finalize_batch_kernel", "taa_resolve_kernel",
                 "ataa_resolve_kernel", "bloom_", "blit_reinhard_kernel"):
        for kd in _find(kernels, part):
            assert kd.get(".vgpr_spill_count", 0) == 0, (part, kd[".name"])
            assert kd[".private_segment_fixed_size"] == 0, (part, kd[".name"], kd[".private_segment_fixed_size"])


def test_f32_fast_marches_keep_their_occupancy(kernels):
    # measured on MI355X (profiles/r01_shader_kernels.jsonl): the packed WGSL march gains 4-7 % from
    # 4 waves/SIMD (<= 128 VGPRs) over 3, the one-ray-per-lane march 3-4 % from 6 (<= 80) over 5 and
    # another 3 % from 7 (<= 72, profiles/r04_ab_wgsl_one_ray_waves.txt).  The GLSL march runs at 8 waves (<= 64
    # VGPRs) since the second half of round 4: a SIMD pairs plain 32-bit VALU operations of two different
    # waves in one quad-cycle and finds a partner more often the more waves it holds (5 -> 6 -> 7 -> 8
    # waves: +2.3 / +5.3 / +6.3 % on the 1080p default preset, profiles/r04_ab_glsl_waves.jsonl); its
    # ~70 B of scratch sit inside the disk / jet sampling branches (two step instances per loop pass since the
    # ping-pong form of round 5), none in a block every pass executes
    for part, limit, scratch in (("wgsl_symplectic_pk_kernel", 128, 0), ("wgsl_symplectic_fast_kernel", 72, 0),
                                 ("glsl_fragment_kernelILi1E", 64, 72)):
        (kd,) = _find(kernels, part)
        assert kd[".vgpr_count"] + kd.get(".agpr_count", 0) <= limit, (part, kd[".vgpr_count"])
        assert kd[".private_segment_fixed_size"] <= scratch, (part, kd[".private_segment_fixed_size"])


def test_glsl_fast_march_keeps_its_scratch_to_the_sampling_bodies(engine_mod):
    # The eight-wave GLSL march spills ~70 B per lane.  Its march loop (two Verlet step instances per pass since
    # round 5: ping-pong positions) may hold only the handful of scratch accesses of the disk / jet sampling
    # bodies and the show_redshift block -- each of them behind a branch a far-field step does not take: the
    # loop head (exit tests + step-size head, up to the first branch INTO a step body's conditional parts)
    # holds none, and every scratch access in the loop has a conditional branch in front of it in its step
    # instance.  (Which blocks a far-field step walks was read off the disassembly: profiles/EXPERIMENTS.md K.)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import isa_histogram as ih
    ins = ih.disassemble(engine_mod.library_path(), "glsl_fragment_kernel<1>")
    assert ins and len(ins) > 3000
    at = {a: i for i, (a, _, _) in enumerate(ins)}

    def target(i):
        off = int(ins[i][2])
        off = off - 65536 if off >= 32768 else off
        return at.get(ins[i][0] + 4 + 4 * off)

    is_br = lambda i: ins[i][1].startswith("s_cbranch") or ins[i][1] == "s_branch"  # noqa: E731
    back = [(target(i), i) for i in range(len(ins)) if is_br(i) and target(i) is not None and target(i) <= i]
    head, latch = max(back, key=lambda b: b[1] - b[0])          # the march loop: the largest backward branch
    assert 1500 < latch - head < 3000
    scratch = [i for i in range(head, latch + 1) if ins[i][1].startswith("scratch_")]
    assert 0 < len(scratch) <= 32, len(scratch)
    # straight-line arithmetic of a step instance: the stretch of >= 60 VALU instructions without a scratch access
    # that starts each instance (acceleration, twist, position update, geometry of the new position)
    runs, cur = [], 0
    for i in range(head, latch + 1):
        if ins[i][1].startswith("scratch_"):
            runs.append(cur)
            cur = 0
        elif ins[i][1].startswith("v_"):
            cur += 1
    runs.append(cur)
    assert sorted(runs)[-2] >= 60, runs     # two such stretches: one per step instance
    # the loop head (exit tests, up to the first conditional branch: it runs on every pass) holds no scratch STORE and at
    # most one reload -- since the jets' dead-shell prefilter (round 6) the allocator parks one loop-invariant dword there
    # (scratch_load_dword at the join, L1-resident); measured with it in place: c2 +1.0 % two streams, +1.3 % one
    # stream, views down the axis x 2.0 ... 2.8 (profiles/r06_ab_glsl_jet_prefilter.jsonl)
    first_br = next(i for i in range(head, latch + 1) if ins[i][1].startswith("s_cbranch"))
    in_head = [ins[i][1] for i in scratch if i < first_br]
    assert len(in_head) <= 1 and all(m.startswith("scratch_load") for m in in_head), in_head


def test_march_kernels_launch_one_wave_blocks(kernels):
    # one-wave blocks for the f32 marches (engine_types.hpp kMarchBlock; 8K x 1024 steps: +10 % measured; short
    # budgets too since the measured-cost dispatch order, profiles/r05_ab_pk_short_one_wave.jsonl)
    for part, size in (("wgsl_symplectic_pk_kernel", 64),
                       ("wgsl_symplectic_fast_kernel", 64), ("wgsl_symplectic_kernel", 64), ("glsl_fragment_kernelILi1E", 64),
                       ("glsl_fragment_kernelILi0E", 64)):
        (kd,) = _find(kernels, part)
        assert kd[".max_flat_workgroup_size"] == size, (part, kd[".max_flat_workgroup_size"])


def test_every_kernel_is_wave64_gfx950(kernels):
    for n, kd in kernels.items():
        assert kd[".wavefront_size"] == 64, n
