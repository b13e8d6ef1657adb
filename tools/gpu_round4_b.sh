#!/bin/bash
# round-4 second pass: the GPU suite again (parity record -> $O), the budget-ray measurement of
# config 4 against the f64 twin, the single-ray entry on the device clocks, the c2 / c5 lines with
# the committed counter passes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r04b}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
GRV_PARITY_JSON=$O/full_frame_parity.json timeout 2700 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
timeout 1200 python tests/measure_c4_budget_rays.py > $O/c4_budget_rays.log 2>&1; echo "budget rays rc=$?"; tail -5 $O/c4_budget_rays.log
timeout 600 python tools/bench_single_ray.py > $O/single_ray_latency.jsonl 2> $O/single_ray.err; echo "single ray rc=$?"; head -8 $O/single_ray_latency.jsonl
timeout 600 python bench.py --config c2 > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"; cut -c1-200 $O/bench_c2.json
timeout 600 python bench.py --config c2 --kernel wgsl > $O/bench_c2wgsl.json 2> $O/bench_c2wgsl.err; echo "c2wgsl rc=$?"; cut -c1-200 $O/bench_c2wgsl.json
