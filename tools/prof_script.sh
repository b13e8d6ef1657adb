#!/bin/bash
# rocprofv3 --kernel-trace --stats of one command, summarised per kernel (run on the GPU box):
#   tools/prof_script.sh <out-name> <command ...>     -> gpurun_out/<out-name>_kernel_stats.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; shift
OUT=$R/gpurun_out/prof_$NAME
rm -rf $OUT; mkdir -p $OUT
CMD="$*"
cd /tmp && export TMPDIR=/tmp
( cd $R && rocprofv3 --kernel-trace --stats -d $OUT -o run -- "$@" > $OUT/stdout.log 2> $OUT/stderr.log )
python - <<PY
import sqlite3, glob, os
dbs = glob.glob(os.path.join("$OUT", "**", "*results.db"), recursive=True)
if not dbs:
    print("$NAME: no rocpd database"); raise SystemExit(1)
def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    depth = 0
    for k, ch in enumerate(n):
        if ch == "<": depth += 1
        elif ch == ">": depth -= 1
        elif ch == "(" and depth == 0: return n[:k]
    return n
con = sqlite3.connect(dbs[0])
rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
lines = ["# rocprofv3 --kernel-trace --stats -- $CMD",
         "%-72s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
for n, c, t, a, p in rows:
    lines.append("%-72s %8d %14.1f %12.2f %7.2f" % (short(n)[:72], c, t, a, p))
reg = con.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, workgroup_x, "
                  "min(grid_x), max(grid_x), count(*), avg(duration), min(duration), max(duration) "
                  "from kernels group by name").fetchall()
lines += ["", "%-52s %5s %5s %5s %7s %7s %5s %10s %10s %6s %12s %12s %12s" % (
    "kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "wg", "grid_min", "grid_max", "n", "avg_ns", "min_ns", "max_ns")]
for r in reg:
    lines.append("%-52s %5d %5d %5d %7d %7d %5d %10d %10d %6d %12.0f %12d %12d" % ((short(r[0])[:52],) + tuple(r[1:])))
open(os.path.join("$R", "gpurun_out", "${NAME}_kernel_stats.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:40]))
PY
rm -rf $OUT/*/  # the rocpd database is scratch; the summary is what travels back
