#!/bin/bash
# round-3 pass on the GPU box: GPU test suite, the bench lines, every rocprofv3 pass of
# tools/profile_gpu.sh (kernel traces, HBM traffic, SQ, the VALU class counters, the VALU
# microbenchmark and its counter calibration).  Outputs under gpurun_out/$1 (default r03c);
# tools/summarize_profiles.py and tools/issue_floor.py turn gpurun_out/prof_$1 into profiles/r03_*.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r03c}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 600 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"; cut -c1-300 $O/bench_c3.json
timeout 600 python bench.py --config c4 --steps 5 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?"; cut -c1-200 $O/bench_c4.json
bash tools/profile_gpu.sh prof_$T > $O/profile_gpu.log 2>&1; tail -3 $O/profile_gpu.log
cat $R/gpurun_out/prof_$T/valu_costs_w4.json | cut -c1-1500
