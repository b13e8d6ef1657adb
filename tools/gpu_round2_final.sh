#!/bin/bash
# round-2 closing pass on the GPU box: full GPU test suite, the bench lines, every rocprofv3 pass
# (tools/profile_gpu.sh), the secondary measurement scripts.  Outputs under gpurun_out/r02f and
# gpurun_out/prof_r02; tools/summarize_profiles.py turns the latter into profiles/r02_*.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02f
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 600 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"; cut -c1-400 $O/bench_c3.json
timeout 600 python bench.py --config c4 --steps 5 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?"; cut -c1-200 $O/bench_c4.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c3_torchrun1.json 2> $O/bench_c3_torchrun1.err; echo "torchrun c3 rc=$?"; cut -c1-200 $O/bench_c3_torchrun1.json
bash tools/profile_gpu.sh prof_r02 > $O/profile_gpu.log 2>&1; tail -3 $O/profile_gpu.log
for s in bench_post bench_shaders bench_renderers bench_single_ray; do
  timeout 600 python tools/$s.py > $O/$s.jsonl 2> $O/$s.err; echo "$s rc=$?"
done
bash tools/prof_script.sh r02_bench_post python tools/bench_post.py > /dev/null 2>&1
bash tools/prof_script.sh r02_bench_shaders python tools/bench_shaders.py > /dev/null 2>&1
bash tools/prof_script.sh r02_bench_renderers python tools/bench_renderers.py > /dev/null 2>&1
cat $O/bench_post.jsonl | cut -c1-200
