#!/bin/bash
# round-2 verification pass: full GPU test suite, the bench lines, rocprofv3 kernel traces of the bench
# commands and of the secondary scripts (summaries land in gpurun_out/, copied into profiles/ by hand)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
timeout 600 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"; cat $O/bench_c3.json
timeout 600 python bench.py --config c4 --steps 5 --warmup 1 > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?"; cat $O/bench_c4.json
timeout 600 python bench.py --arith strict --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_c3_strict.json 2> $O/bench_c3_strict.err; cat $O/bench_c3_strict.json
timeout 300 python tools/bench_post.py 2>/dev/null > $O/post_chain.jsonl; cat $O/post_chain.jsonl
bash tools/prof_script.sh r02_bench_c3 python bench.py --steps 10 --warmup 2 --no-cpu-baseline | head -12
bash tools/prof_script.sh r02_bench_c4 python bench.py --config c4 --steps 5 --warmup 1 --no-cpu-baseline | head -8
bash tools/prof_script.sh r02_bench_c3_strict python bench.py --arith strict --steps 5 --warmup 1 --no-cpu-baseline | head -8
bash tools/prof_script.sh r02_bench_post python tools/bench_post.py | head -14
