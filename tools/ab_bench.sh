#!/bin/bash
# Runs on the GPU box: every ab_libs/lib_*.so takes the in-tree library's place in turn, REPS rounds
# interleaved (A B C A B C ...), bench.py c3 (and c4 with AB_C4=1) each time; one line per run in
# (AB_ARGS="--arith strict" AB_STEPS=8 for the STRICT contract) gpurun_out/$1/ab.jsonl: {"lib", "config", "value", "ms_per_step", "avg_launch_ms"}.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-ab}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
LIB=blackhole-simulation_amd/libgravitas_hip.so
cp $LIB /tmp/lib_orig.so
REPS=${AB_REPS:-3}
for rep in $(seq 1 $REPS); do
  for so in ab_libs/lib_*.so; do
    name=$(basename $so .so)
    cp $so $LIB
    for cfg in c3 ${AB_C4:+c4}; do
      extra=""; [ $cfg = c4 ] && extra="--config c4 --steps 10 --warmup 2"
      [ $cfg = c3 ] && extra="--steps ${AB_STEPS:-20} --warmup 3 ${AB_ARGS:-}"
      timeout ${AB_TIMEOUT:-300} python bench.py $extra --no-cpu-baseline 2> $O/err_${name}_$cfg.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'lib':'$name','config':'$cfg','rep':$rep,'value':d['value'],'ms_per_step':d['ms_per_step'],'avg_launch_ms':d['roofline']['avg_launch_ms']}))" >> $O/ab.jsonl
    done
  done
done
cp /tmp/lib_orig.so $LIB
cat $O/ab.jsonl
