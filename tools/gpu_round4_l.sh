#!/bin/bash
# shader-order f32 kernels in the one-exit loop form: parity (bit for bit vs the oracle) + A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r04l}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_shader_kernels.py tests/test_renderers.py tests/test_golden_shaders.py tests/test_fuzz_parity.py -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
AB_CONFIGS="c2 --arith strict;c4 --arith strict;c2" AB_STEPS=6 bash tools/ab_configs.sh $T > /dev/null
cat $O/ab.jsonl
