cd $GRAFT_REPO_ROOT
export PROFILE_TAG=r06
mkdir -p gpurun_out/r06zu
bash tools/profile_gpu.sh prof_r06zu _c2 _c2wgsl _c4 _c4fast > gpurun_out/r06zu/profile.log 2>&1; tail -2 gpurun_out/r06zu/profile.log
cp gpurun_out/prof_r06zu/summary/traffic.json profiles/traffic.json
for cfg in "c2" "c2 --one-stream" "c2 --kernel wgsl" "c2 --kernel wgsl --one-stream" "c4"; do tag=$(echo $cfg | tr -d ' -'); timeout 900 python bench.py --config $cfg > gpurun_out/r06zu/bench_$tag.json 2> gpurun_out/r06zu/bench_$tag.err; cut -c1-110 gpurun_out/r06zu/bench_$tag.json; done
timeout 900 python tools/camera_sweep.py --configs c2 --out gpurun_out/r06zu/camera_sweep_c2.jsonl > gpurun_out/r06zu/sweep.log 2>&1; wc -l gpurun_out/r06zu/camera_sweep_c2.jsonl
