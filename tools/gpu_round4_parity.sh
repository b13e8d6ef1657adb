#!/bin/bash
# round-4 full-size parity records on the final code objects: config 5 (4K, tol 1e-9), the 8K frame,
# config 4 every pixel (shader order bit for bit; FAST forms: percentiles), renderer FPS.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r04p}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
GRV_PARITY_TOL=1e-9 GRV_PARITY_JSON=$O/full_frame_parity_c5.json timeout 1200 python -m pytest tests/test_full_frame_parity.py -m gpu -q 2>&1 | tail -2
GRV_PARITY_SIZE=7680x4320 GRV_PARITY_JSON=$O/full_frame_parity_8k.json timeout 2400 python -m pytest tests/test_full_frame_parity.py -m gpu -q 2>&1 | tail -2
GRV_C4_STRIDE=1 GRV_C4_JSON=$O/full_frame_parity_c4.jsonl timeout 2400 python -m pytest tests/test_shader_kernels.py -m gpu -q -k test_config4_bench_form 2>&1 | tail -2
timeout 600 python tools/bench_renderers.py > $O/renderers.jsonl 2> $O/renderers.err; cat $O/renderers.jsonl | cut -c1-200
timeout 600 python tools/bench_shaders.py > $O/shader_kernels.jsonl 2> $O/shaders.err; cut -c1-160 $O/shader_kernels.jsonl
