#!/bin/bash
# round-4 closing pass on the GPU box: smoke(), the whole GPU suite (full-frame parity record), the
# bench lines of every config, the budget-ray measurement, the rocprofv3 passes of the f32 kernels
# (all six dominant kernels: the code hashes stamp profiles/traffic.json), a 100-seed randomised campaign.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r04f}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
GRV_C2_JSON=$O/full_frame_parity_c2.jsonl GRV_PARITY_JSON=$O/full_frame_parity.json timeout 2700 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log
for cfg in "c3" "c2" "c2 --kernel wgsl" "c4" "c5 --steps 5 --warmup 1" "c2 --one-stream" "c2 --kernel wgsl --one-stream" "c2 --arith strict --no-cpu-baseline"; do
  tag=$(echo $cfg | tr -d ' -' ); timeout 900 python bench.py --config $cfg > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "$cfg rc=$?"; cut -c1-120 $O/bench_$tag.json
done
timeout 1200 python tests/measure_c4_budget_rays.py > $O/c4_budget_rays.log 2>&1; echo "budget rays rc=$?"
( GRV_FUZZ_SEEDS=${FUZZ_SEEDS:-100} timeout 2400 python -m pytest tests/test_fuzz_parity.py -m gpu -q -x 2>&1 | tail -3 ) > $O/fuzz.log 2>&1; cat $O/fuzz.log
bash tools/profile_gpu.sh prof_$T base _strict _c4 _c4fast _c2 _c2wgsl _c5 > $O/profile_gpu.log 2>&1; tail -2 $O/profile_gpu.log
