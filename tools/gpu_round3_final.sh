#!/bin/bash
# round-3 closing pass on the GPU box: smoke(), the whole GPU suite, the bench lines, every rocprofv3
# pass (tools/profile_gpu.sh), a 200-seed randomised campaign on the final kernels.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r03z}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 600 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"; cut -c1-200 $O/bench_c3.json
timeout 600 python bench.py --config c4 --steps 10 --warmup 2 > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?"; cut -c1-200 $O/bench_c4.json
bash tools/profile_gpu.sh prof_$T > $O/profile_gpu.log 2>&1; tail -2 $O/profile_gpu.log
( GRV_FUZZ_SEEDS=200 timeout 2400 python -m pytest tests/test_fuzz_parity.py -m gpu -q -x 2>&1 | tail -3 ) > $O/fuzz200.log 2>&1; cat $O/fuzz200.log
