#!/bin/bash
# rocprofv3 kernel-trace of the secondary measurement scripts (run on the GPU box through gpurun).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for s in bench_shaders bench_post bench_renderers; do
  rocprofv3 --kernel-trace --stats -d $OUT/$s -o run -- python $R/tools/$s.py > $OUT/$s.out 2> $OUT/$s.err
done
python - <<PY
import sqlite3, glob, os
out = "$OUT"
for s in ("bench_shaders", "bench_post", "bench_renderers"):
    dbs = glob.glob(os.path.join(out, s, "**", "*results.db"), recursive=True)
    if not dbs:
        print(s, "no db"); continue
    rows = sqlite3.connect(dbs[0]).execute(
        "select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    lines = ["# rocprofv3 --kernel-trace --stats -- python tools/%s.py" % s,
             "%-70s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for n, c, t, a, p in rows:
        n = n.replace("void ", "").replace("(anonymous namespace)::", "")
        n = n.split("(")[0] if "<" not in n.split("(")[0] else n[:n.rfind(">") + 1] if ">" in n else n
        lines.append("%-70s %8d %14.1f %12.2f %7.2f" % (n[:70], c, t, a, p))
    open(os.path.join("$R", "gpurun_out", "r01_%s_kernel_stats.txt" % s), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:12]))
PY
