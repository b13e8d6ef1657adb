cd $GRAFT_REPO_ROOT
export PROFILE_TAG=r06
mkdir -p gpurun_out/r06zz
GRV_C2_JSON=gpurun_out/r06zz/full_frame_parity_c2.jsonl timeout 900 python -m pytest tests/test_shader_kernels.py -m gpu -q -k test_config2_bench_form 2>&1 | tail -1
GRV_C4_STRIDE=1 GRV_C4_JSON=gpurun_out/r06zz/full_frame_parity_c4.jsonl timeout 2400 python -m pytest tests/test_shader_kernels.py -m gpu -q -k test_config4_bench_form 2>&1 | tail -1
bash tools/gpu_session.sh r06zz fuzzfast:500 2>&1 | tail -4
