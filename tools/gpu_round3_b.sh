#!/bin/bash
# round-3 pass B: the new path tests + the addon tests, then every rocprofv3 pass (tools/profile_gpu.sh)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r03d}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_paths.py tests/test_napi_addon.py tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
bash tools/profile_gpu.sh prof_$T > $O/profile_gpu.log 2>&1; tail -3 $O/profile_gpu.log
