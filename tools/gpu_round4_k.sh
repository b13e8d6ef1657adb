#!/bin/bash
# A/B of the one-ray FAST WGSL march's loop forms + bitwise comparison of their images (one box)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r04k}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
LIB=blackhole-simulation_amd/libgravitas_hip.so
cp $LIB /tmp/lib_orig.so
for so in ab_libs/lib_*.so; do name=$(basename $so .so); cp $so $LIB; AB_KERNEL=wgsl python tools/ab_glsl_identical.py /tmp/$name.npz 2>/dev/null; done
ls /tmp/lib_*.npz | xargs python tools/ab_glsl_identical.py | tee $O/identical.txt
cp /tmp/lib_orig.so $LIB
AB_CONFIGS="c4 --arith fast;c2 --kernel wgsl --arith fast --one-stream" AB_STEPS=8 bash tools/ab_configs.sh $T > /dev/null
cat $O/ab.jsonl
timeout 1200 python -m pytest tests/test_shader_kernels.py tests/test_renderers.py -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
