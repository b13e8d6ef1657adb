#!/usr/bin/env python3
"""Distributions of every figure in a GRV_FUZZ_REPORT file (tests/test_fuzz_parity.py writes one JSON row per
configuration): per test kind, min / p1 / p50 / p99 / max of each numeric column.
usage: tools/fuzz_report_summary.py gpurun_out/<tag>/fuzz_fast_report.jsonl"""
import json
import sys

import numpy as np

SETUP = {"seed", "kind", "mass", "spin", "method", "tolerance", "initial_step", "max_steps", "escape_radius",
         "renormalize_interval", "W", "H", "r0", "theta", "fovy", "ray", "frame", "width", "height"}


def main(path):
    rows = {}
    for line in open(path):
        line = line.strip()
        if line:
            d = json.loads(line)
            rows.setdefault(d.pop("test"), []).append(d)
    print("records:", json.dumps({k: len(v) for k, v in rows.items()}))
    for test, rs in rows.items():
        print(f"\n## {test}: {len(rs)} records")
        cols = sorted({k for r in rs for k, v in r.items() if isinstance(v, (int, float)) and not isinstance(v, bool)} - SETUP)
        for c in cols:
            x = np.array([r[c] for r in rs if c in r and r[c] is not None], dtype=np.float64)
            x = x[np.isfinite(x)]
            if x.size:
                q = np.percentile(x, [1, 50, 99])
                print(f"  {c:28s} min {x.min():.6g}  p1 {q[0]:.6g}  p50 {q[1]:.6g}  p99 {q[2]:.6g}  max {x.max():.6g}")


if __name__ == "__main__":
    main(sys.argv[1])
