#!/bin/bash
# round-4 re-stamp after the Horner-in-SGPR change of the FAST f64 right-hand side (bits unchanged): the
# whole GPU suite with the full-frame record, the c3 bench line, the 8K frame, the rocprofv3 passes of
# integrate_segment_kernel<1,1,0> only (the other kernels' passes of tools/gpu_round4_f.sh stay valid:
# their code hashes did not move).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r04h}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
GRV_PARITY_JSON=$O/full_frame_parity.json timeout 2700 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 900 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"; cut -c1-120 $O/bench_c3.json
GRV_PARITY_SIZE=7680x4320 GRV_PARITY_JSON=$O/full_frame_parity_8k.json timeout 2400 python -m pytest tests/test_full_frame_parity.py -m gpu -q 2>&1 | tail -2
bash tools/profile_gpu.sh prof_$T base > $O/profile_gpu.log 2>&1; tail -2 $O/profile_gpu.log
