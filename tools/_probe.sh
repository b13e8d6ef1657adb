cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06e
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "test_gpu_images or test_napi_frames or test_napi_addon" 2>&1 | tail -3
for i in 1 2; do
node napi/bench_frames.js --config c3 --form device | cut -c1-120
node napi/bench_frames.js --config c3 --form async | cut -c1-120
node napi/bench_frames.js --config c3 --form read | cut -c1-120
node napi/bench_frames.js --config c2 --form device | cut -c1-120
node napi/bench_frames.js --config c2 --form read | cut -c1-120
done
rocprofv3 --kernel-trace --memory-copy-trace -d $O/prof_async -o async -- node napi/bench_frames.js --config c3 --form async --steps 10 > $O/prof_async.log 2>&1
rocprofv3 --kernel-trace --memory-copy-trace -d $O/prof_c2read -o c2read -- node napi/bench_frames.js --config c2 --form read --steps 40 --warmup 10 > $O/prof_c2read.log 2>&1
find $O -name "*.db" -exec sh -c 'mv {} '$O'/$(basename {})' \;
