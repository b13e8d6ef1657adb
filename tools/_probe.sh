cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06w
timeout 300 node napi/control_latency.js gpurun_out/r06w/control_latency.json | cut -c1-1500
timeout 1500 python -m pytest tests -m gpu -q -x -k "test_gpu_images or test_gpu_async or test_gpu_parity or test_gpu_paths or test_napi_addon or test_gpu_lifecycle or test_fuzz" 2>&1 | tail -3
