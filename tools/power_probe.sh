#!/bin/bash
# Samples the GPU's clock and package power (rocm-smi) while bench.py runs a config for a few seconds.
# usage (GPU box): tools/power_probe.sh <out.txt> <bench args...>     e.g. tools/power_probe.sh out.txt --config c3
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
{ echo "# idle"; rocm-smi --showclocks --showpower --showmaxpower 2>/dev/null | grep -i "sclk\|power\|mclk" ; } > $OUT
timeout 300 python bench.py "$@" --steps ${PROBE_STEPS:-300} --warmup 3 --no-cpu-baseline > $OUT.bench.json 2> $OUT.err &
BP=$!
sleep ${PROBE_DELAY:-25}   # import + engine creation + warm-up
for i in 1 2 3 4 5 6; do
  kill -0 $BP 2>/dev/null || break
  { echo "# sample $i under: bench.py $*"; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" ; } >> $OUT
  sleep 1
done
wait $BP
cut -c1-200 $OUT.bench.json >> $OUT
