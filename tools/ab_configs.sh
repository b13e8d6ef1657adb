#!/bin/bash
# Runs on the GPU box: every ab_libs/lib_*.so takes the in-tree library's place in turn, REPS rounds
# interleaved, under `bench.py --config X` for every X of AB_CONFIGS (default "c2;c2 --kernel wgsl;c3;c4").
# One JSON line per run in gpurun_out/$1/ab.jsonl.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-abc}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
LIB=blackhole-simulation_amd/libgravitas_hip.so
cp $LIB /tmp/lib_orig.so
IFS=';' read -ra CFGS <<< "${AB_CONFIGS:-c2;c2 --kernel wgsl;c3;c4}"
for rep in $(seq 1 ${AB_REPS:-3}); do
  for so in ab_libs/lib_*.so; do
    name=$(basename $so .so); cp $so $LIB
    for cfg in "${CFGS[@]}"; do
      sw="--steps ${AB_STEPS:-10} --warmup 2"
      case "$cfg" in c2*) sw="--steps ${AB_STEPS_C2:-300} --warmup 60";; esac  # 1-2 ms frames: a short window sits on the clock ramp
      timeout ${AB_TIMEOUT:-240} python bench.py --config $cfg $sw --no-cpu-baseline 2> $O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'lib':'$name','config':'$cfg','rep':$rep,'value':d['value'],'ms_per_step':d['ms_per_step'],'avg_launch_ms':d['roofline']['avg_launch_ms'],'steps_per_frame':d['config']['accepted_steps_per_frame']}))" >> $O/ab.jsonl
    done
  done
done
cp /tmp/lib_orig.so $LIB
cat $O/ab.jsonl
