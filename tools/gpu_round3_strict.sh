#!/bin/bash
# Round 3, STRICT division forms: the strict-math checks, the STRICT parity tests, the whole-frame
# STRICT parity (0 bits) and the A/B of ab_libs/lib_s_*.so under bench.py --arith strict on one box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r03s2}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
python -m pytest tests/test_ref_libm.py -m gpu -x -q > $O/libm.log 2>&1; echo "libm rc=$?" | tee -a $O/summary.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py tests/test_fuzz_parity.py -m gpu -x -q > $O/parity.log 2>&1; echo "parity rc=$?" | tee -a $O/summary.txt
GRV_PARITY_JSON=$O/full_frame_parity.json python -m pytest tests/test_full_frame_parity.py -m gpu -x -q > $O/full.log 2>&1; echo "full rc=$?" | tee -a $O/summary.txt
AB_ARGS="--arith strict" AB_STEPS=6 AB_REPS=2 bash tools/ab_bench.sh $T > $O/ab.log 2>&1
tail -3 $O/libm.log $O/parity.log $O/full.log
cat $O/ab.jsonl
