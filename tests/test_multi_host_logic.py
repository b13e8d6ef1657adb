"""The multi-GPU host logic (csrc/multi_core.hpp: one worker thread per rank, the exchange buffers and events of
both frame parities, the frame skeleton of grv_render_frame_multi*) runs here on the CPU over a mock Api whose
streams are real threads and whose copies touch real memory (tests/host/multi_tsan.cpp): G = 2, 4, 8 ranks x
{peer copy, RCCL-shaped group} x {RGBA32F, RGBA16F}, frames of both parities in flight, buffer regrowth, and the
three injected faults -- every assembled image compared pixel by pixel.  The same program under ThreadSanitizer
is a leg of oracle/sanitize_host.sh (SURVEY section 5 names TSan; the pool has no second device to race on).

The two mutants prove the harness can see what it is for: without the wait that frees a parity's receive slots,
or without the wait for the ranks' arrivals, images come out wrong."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "multi_tsan.cpp")
CXX = shutil.which("g++") or shutil.which("clang++")

pytestmark = pytest.mark.skipif(CXX is None, reason="no host C++ compiler")


def _build(tmp_path, name, *defs):
    exe = tmp_path / name
    r = subprocess.run([CXX, "-O2", "-std=c++17", "-pthread", "-Wall", "-Wextra", "-Werror", *["-D" + d for d in defs], SRC,
                        "-o", str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return str(exe)


def test_frame_skeleton_assembles_every_frame_over_the_mock_api(tmp_path):
    exe = _build(tmp_path, "multi_plain")
    r = subprocess.run([exe, "160"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["failed"] == 0 and res["bad_pixels"] == 0 and res["combinations"] == 15
    assert res["frames_checked"] >= 12 * 140  # (the frames that carried an injected fault are not images)


@pytest.mark.parametrize("mutant", ["GRVMULTI_MUTANT_NO_SLOT_WAIT", "GRVMULTI_MUTANT_NO_ARRIVED_WAIT"])
def test_harness_notices_a_missing_event(tmp_path, mutant):
    exe = _build(tmp_path, "multi_" + mutant.lower(), mutant)
    r = subprocess.run([exe, "120"], capture_output=True, text=True, timeout=600)
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode != 0 and res["bad_pixels"] > 0, res
