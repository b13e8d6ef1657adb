#!/bin/bash
# round-4 third pass: A/B of the FAST f32 marches (ab_libs/lib_base.so: general noise path + the
# shader's literal r < r_stop exit; lib_new.so: lattice noise + NaN-proof exit) under bench.py c2 / c4
# on ONE box, interleaved; then the suite and the budget-ray measurement on the new library.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r04c}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
LIB=blackhole-simulation_amd/libgravitas_hip.so
cp $LIB /tmp/lib_orig.so
for rep in 1 2 3; do
  for so in ab_libs/lib_base.so ab_libs/lib_new.so; do
    name=$(basename $so .so); cp $so $LIB
    for cfg in "c2" "c2 --kernel wgsl" "c4"; do
      python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline 2> $O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'lib':'$name','config':'$cfg','rep':$rep,'value':d['value'],'ms_per_step':d['ms_per_step'],'avg_launch_ms':d['roofline']['avg_launch_ms'],'steps_per_frame':d['config']['accepted_steps_per_frame']}))" >> $O/ab.jsonl
    done
  done
done
cp /tmp/lib_orig.so $LIB
cat $O/ab.jsonl
timeout 2700 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 1200 python tests/measure_c4_budget_rays.py > $O/c4_budget_rays.log 2>&1; echo "budget rays rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r04_c4_budget_rays.json')); print(d['rays_at_budget']); A=d['A_pixels_where_any_form_reaches_the_budget']; print(A['pixels'],A['f64_at_budget'],{k:(v['at_budget'],v['agrees_with_f64_on_budget']) for k,v in A['forms'].items()}); print({k:(v['pixels'],v['form_closer_to_f64'],v['shader_order_closer_to_f64']) for k,v in d['B_pixels_beyond_5e-2_of_peak'].items()})"
