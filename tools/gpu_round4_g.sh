#!/bin/bash
# A/B of GLSL FAST loop forms + bitwise comparison of their images (one box)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r04g}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
LIB=blackhole-simulation_amd/libgravitas_hip.so
cp $LIB /tmp/lib_orig.so
for so in ab_libs/lib_*.so; do name=$(basename $so .so); cp $so $LIB; python tools/ab_glsl_identical.py /tmp/$name.npz 2>/dev/null; done
ls /tmp/lib_*.npz | xargs python tools/ab_glsl_identical.py | tee $O/identical.txt
for rep in 1 2 3; do
  for so in ab_libs/lib_*.so; do
    name=$(basename $so .so); cp $so $LIB
    python bench.py --config c2 --steps 20 --warmup 3 --no-cpu-baseline 2> $O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'lib':'$name','rep':$rep,'value':d['value'],'ms_per_step':d['ms_per_step'],'avg_launch_ms':d['roofline']['avg_launch_ms']}))" >> $O/ab.jsonl
  done
done
cp /tmp/lib_orig.so $LIB
cat $O/ab.jsonl
timeout 1200 python -m pytest tests/test_shader_kernels.py tests/test_renderers.py tests/test_golden_shaders.py -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
