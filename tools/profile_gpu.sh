#!/bin/bash
# rocprofv3 passes for the bench workloads (run on the GPU box through gpurun).  Kernel-trace/stats
# and each PMC set are separate runs, as the guide prescribes.  Passes per dominant kernel:
#   ""        bench.py                 integrate_segment_kernel<1,1,0>  (FAST f64, the bench line)
#   _strict   bench.py --arith strict  integrate_segment_kernel<1,0,0>  (reference order, the FFI's contract)
#   _c4       bench.py --config c4     wgsl_symplectic_pk_kernel        (f32 march of the scaling config, two rays per lane)
#   _c4fast   bench.py --config c4 --arith fast   wgsl_symplectic_fast_kernel  (one ray per lane)
# code_hashes.json stamps the passes with the code objects of the library that ran them
# (tools/summarize_profiles.py -> profiles/traffic.json; bench.py drops figures whose stamp differs).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-prof}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python - <<PY
import json, sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tools")
import kernel_resources as kr
lib = "$R/blackhole-simulation_amd/libgravitas_hip.so"
names = ["integrate_segment_kernel<1,1,0>", "integrate_segment_kernel<1,0,0>", "wgsl_symplectic_fast_kernel",
         "wgsl_symplectic_pk_kernel"]
json.dump({n: kr.kernel_code_hash(lib, n) for n in names}, open("$OUT/code_hashes.json", "w"), indent=1)
PY
run() { # label, rocprof args..., -- bench args
  local label=$1; shift
  local rp=(); while [ "$1" != "--" ]; do rp+=("$1"); shift; done; shift
  ( cd $R && rocprofv3 "${rp[@]}" -d $OUT/$label -o bench -- python bench.py "$@" --no-cpu-baseline > $OUT/${label}_bench.json 2> $OUT/$label.err )
  # rocprofv3 nests the database under <host>/: lift it to where summarize_profiles.py looks
  db=$(find $OUT/$label -name "bench_results.db" | head -1); [ -n "$db" ] && [ "$db" != "$OUT/$label/bench_results.db" ] && mv "$db" $OUT/$label/bench_results.db
}
run trace        --kernel-trace --stats -- --steps 10 --warmup 2
run trace_k16    --kernel-trace --stats -- --steps 10 --warmup 2 --segment-tries 16
run trace_strict --kernel-trace --stats -- --steps 5 --warmup 1 --arith strict
run trace_c4     --kernel-trace --stats -- --steps 5 --warmup 1 --config c4
SQ="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
for sfx in "" _strict _c4 _c4fast; do
  case "$sfx" in "") extra="";; _strict) extra="--arith strict";; _c4) extra="--config c4";; _c4fast) extra="--config c4 --arith fast";; esac
  run pmc_fetch$sfx --pmc FETCH_SIZE -- --steps 2 --warmup 1 $extra
  run pmc_write$sfx --pmc WRITE_SIZE -- --steps 2 --warmup 1 $extra
  run pmc_sq$sfx    --pmc $SQ -- --steps 2 --warmup 1 $extra
done
# dynamic VALU mix by hardware class counter (tools/issue_floor.py prices it with the microbenchmark's costs)
CLS32="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64"
# flops as the hardware counts them, and the issue-occupancy pair of one and the same run
OCC="SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_FLOPS_FP32_TRANS SQ_INSTS_VALU_FLOPS_FP64_TRANS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 GRBM_GUI_ACTIVE"
CLS64="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU"
for sfx in "" _strict _c4 _c4fast; do
  case "$sfx" in "") extra="";; _strict) extra="--arith strict";; _c4) extra="--config c4";; _c4fast) extra="--config c4 --arith fast";; esac
  run pmc_cls32$sfx --pmc $CLS32 -- --steps 2 --warmup 1 $extra
  run pmc_cls64$sfx --pmc $CLS64 -- --steps 2 --warmup 1 $extra
  run pmc_occ$sfx --pmc $OCC -- --steps 2 --warmup 1 $extra
done
# the microbenchmark: costs in shader cycles (plain runs), and the same binary under the class counters
# (which counter does a mnemonic land in, what do SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES say per instruction)
if hipcc -O2 --offload-arch=gfx950 $R/tools/valu_microbench.hip -o /tmp/valu_microbench 2> $OUT/valu_microbench_build.err; then
  for w in 1 2 4 8; do /tmp/valu_microbench $w > $OUT/valu_costs_w$w.json 2>> $OUT/valu_microbench.err; done
  mb() { local label=$1; shift
    rocprofv3 "$@" -d $OUT/$label -o bench -- /tmp/valu_microbench 4 > /dev/null 2> $OUT/$label.err
    db=$(find $OUT/$label -name "bench_results.db" | head -1); [ -n "$db" ] && [ "$db" != "$OUT/$label/bench_results.db" ] && mv "$db" $OUT/$label/bench_results.db; }
  mb mb_cls32 --pmc $CLS32
  mb mb_cls64 --pmc $CLS64
  mb mb_sq --pmc $SQ
  mb mb_occ --pmc $OCC
fi
run pmc_fetch_k16 --pmc FETCH_SIZE -- --steps 2 --warmup 1 --segment-tries 16
run pmc_write_k16 --pmc WRITE_SIZE -- --steps 2 --warmup 1 --segment-tries 16
find $OUT -name "*.csv" -delete
find $OUT -name "*.db" | head -40
du -sh $OUT
