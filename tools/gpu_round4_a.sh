#!/bin/bash
# round-4 first pass on the GPU box: smoke(), the whole GPU suite (the full-frame parity record goes
# to $O/full_frame_parity.json), the bench lines incl. the new --config c2 (both kernels) and a
# re-stamped --config c5, then the rocprofv3 passes of the c2 kernels and the c5 trace.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r04a}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
GRV_PARITY_JSON=$O/full_frame_parity.json timeout 2700 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
timeout 600 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"; cut -c1-200 $O/bench_c3.json
timeout 600 python bench.py --config c2 > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"; cut -c1-400 $O/bench_c2.json; tail -3 $O/bench_c2.err
timeout 600 python bench.py --config c2 --kernel wgsl > $O/bench_c2wgsl.json 2> $O/bench_c2wgsl.err; echo "c2wgsl rc=$?"; cut -c1-300 $O/bench_c2wgsl.json
timeout 600 python bench.py --config c2 --arith strict --no-cpu-baseline > $O/bench_c2strict.json 2> $O/bench_c2strict.err; echo "c2strict rc=$?"; cut -c1-200 $O/bench_c2strict.json
timeout 900 python bench.py --config c5 --steps 5 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"; cut -c1-200 $O/bench_c5.json
timeout 300 python bench.py --gpus 2; echo "bare --gpus 2 on this box: rc=$?"
bash tools/profile_gpu.sh prof_$T _c2 _c2wgsl _c5 > $O/profile_gpu.log 2>&1; tail -3 $O/profile_gpu.log
