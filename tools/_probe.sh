cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06zq
for cfg in "c2" "c2 --one-stream"; do tag=$(echo $cfg | tr -d ' -'); timeout 900 python bench.py --config $cfg > gpurun_out/r06zq/bench_$tag.json 2> gpurun_out/r06zq/bench_$tag.err; cut -c1-110 gpurun_out/r06zq/bench_$tag.json; done
timeout 900 python tools/camera_sweep.py --configs c2 --out gpurun_out/r06zq/camera_sweep_c2.jsonl > gpurun_out/r06zq/sweep.log 2>&1; wc -l gpurun_out/r06zq/camera_sweep_c2.jsonl
timeout 600 node napi/bench_frames.js --config c2 --form device > gpurun_out/r06zq/napi_c2.json 2> gpurun_out/r06zq/napi.err; tail -c 400 gpurun_out/r06zq/napi_c2.json
