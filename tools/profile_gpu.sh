#!/bin/bash
# rocprofv3 passes for the bench workload (run on the GPU box through gpurun).
# kernel-trace/stats and each PMC set are separate runs, as the guide prescribes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $R/bench.py $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --stats -d $OUT/trace_k16 -o bench -- python $R/bench.py $ARGS --segment-tries 16 > $OUT/trace_k16_bench.json 2> $OUT/trace_k16.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_k16 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --segment-tries 16 > /dev/null 2> $OUT/pmc_fetch_k16.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_k16 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --segment-tries 16 > /dev/null 2> $OUT/pmc_write_k16.err
find $OUT -name "*.csv" | head -50
du -sh $OUT
