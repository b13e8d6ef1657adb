#!/usr/bin/env python3
"""Assembles profiles/rNN_napi_frames.json from one `tools/gpu_session.sh <tag> frames` session: bench.py's line of
c3 and c2 (Python ctypes over the C ABI) beside napi/bench_frames.js's lines of the same configs on the same box
(JavaScript through the N-API addon over the same C ABI), with the ratios the record is read for.
usage: tools/assemble_napi_frames.py gpurun_out/<tag> > profiles/r06_napi_frames.json"""
import json
import os
import sys


def main():
    d = sys.argv[1]
    out = {"session": os.path.basename(d.rstrip("/")), "configs": {}}
    for cfg in ("c3", "c2", "c4", "c5"):
        if not os.path.exists(os.path.join(d, "napi_frames_%s.jsonl" % cfg)):
            continue
        py = json.loads(open(os.path.join(d, "frames_benchpy_%s.json" % cfg)).read().strip().splitlines()[-1])
        js = [json.loads(l) for l in open(os.path.join(d, "napi_frames_%s.jsonl" % cfg)) if l.strip()]
        rec = {"bench_py": {k: py[k] for k in ("value", "ms_per_step", "steps", "warmup")},
               "bench_py_accepted_steps_per_frame": py["config"]["accepted_steps_per_frame"],
               "bench_py_frames_in_flight": py["config"].get("frames_in_flight"), "node": {}}
        for l in js:
            rec["node"][l["form"]] = {"value": l["value"], "ms_per_step": l["ms_per_step"], "steps": l["steps"],
                                      "vs_bench_py": round(l["value"] / py["value"], 4),
                                      "extra_ms_per_frame_vs_device_form": None,
                                      "accepted_steps_per_frame": l["config"]["accepted_steps_per_frame"],
                                      "pixels": l.get("pixels"), "d2h_bytes_per_frame": l.get("d2h_bytes_per_frame", 0),
                                      **({"host_queue_ms_per_frame": l["host_queue_ms_per_frame"]} if "host_queue_ms_per_frame" in l else {})}
        dev = rec["node"].get("device")
        for f, r in rec["node"].items():
            r["extra_ms_per_frame_vs_device_form"] = round(r["ms_per_step"] - dev["ms_per_step"], 4) if dev else None
        rec["host"] = js[0]["host"] if js else None
        rec["workload"] = js[0]["config"]["workload"] if js else None
        out["configs"][cfg] = rec
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
