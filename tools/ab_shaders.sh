#!/bin/bash
# Runs on the GPU box: every ab_libs/lib_*.so in turn under tools/bench_shaders.py (the f32 marches),
# FAST cases only: gpurun_out/$1/ab_shaders.txt
set -u
cd ${GRAFT_REPO_ROOT:-.}
T=${1:-absh}
LIB=blackhole-simulation_amd/libgravitas_hip.so
cp $LIB /tmp/lib_orig.so
mkdir -p gpurun_out/$T
for rep in 1 2; do for so in ab_libs/lib_*.so; do n=$(basename $so .so); cp $so $LIB; python tools/bench_shaders.py 2>/dev/null | grep -i "FAST" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$n', d['case'][:56], d['width'], d['ms_per_frame'], d['Mray_steps_per_s'])
" | tee -a gpurun_out/$T/ab_shaders.txt; done; done
cp /tmp/lib_orig.so $LIB
