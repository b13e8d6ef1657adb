#!/bin/bash
# round-3 secondary measurements on the GPU box with the final kernels: full-frame parity at 4K and 8K
# (STRICT bit for bit, FAST percentiles), the f32 FAST tail, the secondary bench scripts, RCCL / native
# one-rank walks of bench.py.  Outputs under gpurun_out/$1; copied to profiles/r03_* afterwards.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r03s}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
GRV_PARITY_JSON=$O/full_frame_parity.json timeout 900 python -m pytest tests/test_full_frame_parity.py -m gpu -q > $O/parity4k.log 2>&1; tail -2 $O/parity4k.log
GRV_PARITY_SIZE=7680x4320 GRV_PARITY_JSON=$O/full_frame_parity_8k.json timeout 1500 python -m pytest tests/test_full_frame_parity.py -m gpu -q > $O/parity8k.log 2>&1; tail -2 $O/parity8k.log
timeout 900 python tests/measure_f32_fast_tail.py > $O/f32_tail.log 2>&1; tail -2 $O/f32_tail.log; cp $R/gpurun_out/r03_f32_fast_tail.json $O/ 2>/dev/null
for s in bench_shaders bench_post bench_renderers bench_single_ray bench_batch; do
  timeout 600 python tools/$s.py > $O/$s.jsonl 2> $O/$s.err; echo "$s rc=$?"
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c3_torchrun1.json 2> $O/bench_c3_torchrun1.err; echo "torchrun c3 rc=$?"
timeout 600 python bench.py --native --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c3_native1.json 2> $O/bench_c3_native1.err; echo "native c3 rc=$?"
GRV_BENCH_ONE_DEVICE=1 GRV_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_c3_gloo2.json 2> $O/bench_c3_gloo2.err; echo "bare --gpus 2 rc=$?"; cut -c1-200 $O/bench_c3_gloo2.json
cat $O/full_frame_parity.json | head -60
