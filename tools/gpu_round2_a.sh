#!/bin/bash
# round-2 first GPU pass: full GPU test suite, bench lines (c3, c4, 1-rank torchrun walk), rank shares, single ray
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
cat $O/bench_c3.json
timeout 600 python bench.py --config c4 --steps 5 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?"
cat $O/bench_c4.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c3_torchrun1.json 2> $O/bench_c3_torchrun1.err; echo "torchrun c3 rc=$?"
cat $O/bench_c3_torchrun1.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 1 --config c4 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4_torchrun1.json 2> $O/bench_c4_torchrun1.err; echo "torchrun c4 rc=$?"
cat $O/bench_c4_torchrun1.json
timeout 600 python bench.py --arith strict --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_c3_strict.json 2> $O/bench_c3_strict.err
cat $O/bench_c3_strict.json
timeout 600 python tools/bench_rank_share.py c3 > $O/rank_share_c3.jsonl 2> $O/rank_share_c3.err; cat $O/rank_share_c3.jsonl
timeout 600 python tools/bench_rank_share.py c4 > $O/rank_share_c4.jsonl 2> $O/rank_share_c4.err; cat $O/rank_share_c4.jsonl
timeout 300 python tools/bench_single_ray.py > $O/single_ray.jsonl 2> $O/single_ray.err; head -5 $O/single_ray.jsonl
