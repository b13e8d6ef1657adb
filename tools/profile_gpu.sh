#!/bin/bash
# rocprofv3 passes for the bench workloads (run on the GPU box through gpurun).  Kernel-trace/stats
# and each PMC set are separate runs, as the guide prescribes.  Passes per dominant kernel:
#   ""        bench.py                 integrate_segment_kernel<1,1,0>  (FAST f64, the bench line)
#   _strict   bench.py --arith strict  integrate_segment_kernel<1,0,0>  (reference order, the FFI's contract)
#   _c4       bench.py --config c4     wgsl_symplectic_pk_kernel        (f32 march of the scaling config, two rays per lane)
#   _c4fast   bench.py --config c4 --arith fast   wgsl_symplectic_fast_kernel  (one ray per lane)
#   _c2       bench.py --config c2     glsl_fragment_kernel<1>          (BASELINE configs[1]: the WebGL shader's Verlet march, FAST)
#   _c2wgsl   bench.py --config c2 --kernel wgsl   wgsl_symplectic_pk_kernel at 1920x1080 / 512 steps (filed as wgsl_symplectic_pk_kernel@1920x1080)
#   _c5       bench.py --config c5     integrate_segment_kernel<1,0,0> at tol 1e-9 (trace only; the PMC passes are _strict's)
# usage: tools/profile_gpu.sh <out dir under gpurun_out> [suffixes...]   (default: every suffix; "base" = the "" passes)
# PROFILE_TAG (default r05) names the records; $OUT/summary holds what is copied into profiles/ afterwards.
# code_hashes.json stamps the passes with the code objects of the library that ran them
# (tools/summarize_profiles.py -> profiles/traffic.json; bench.py drops figures whose stamp differs).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-prof}
shift || true
SUFFIXES=("$@"); [ ${#SUFFIXES[@]} -eq 0 ] && SUFFIXES=(base _strict _c4 _c4fast _c2 _c2wgsl k16 microbench)
want() { local s; for s in "${SUFFIXES[@]}"; do [ "$s" == "$1" ] && return 0; done; return 1; }
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python - <<PY
import json, sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tools")
import kernel_resources as kr
lib = "$R/blackhole-simulation_amd/libgravitas_hip.so"
names = ["integrate_segment_kernel<1,1,0>", "integrate_segment_kernel<1,0,0>", "wgsl_symplectic_fast_kernel",
         "wgsl_symplectic_pk_kernel", "glsl_fragment_kernel<1>", "glsl_fragment_kernel<0>"]
json.dump({n: kr.kernel_code_hash(lib, n) for n in names}, open("$OUT/code_hashes.json", "w"), indent=1)
PY
run() { # label, rocprof args..., -- bench args
  local label=$1; shift
  local rp=(); while [ "$1" != "--" ]; do rp+=("$1"); shift; done; shift
  ( cd $R && timeout 600 rocprofv3 "${rp[@]}" -d $OUT/$label -o bench -- python bench.py "$@" --no-cpu-baseline > $OUT/${label}_bench.json 2> $OUT/$label.err )
  # rocprofv3 nests the database under <host>/: lift it to where summarize_profiles.py looks
  db=$(find $OUT/$label -name "bench_results.db" | head -1); [ -n "$db" ] && [ "$db" != "$OUT/$label/bench_results.db" ] && mv "$db" $OUT/$label/bench_results.db
}
extra_of() { case "$1" in base) echo "";; _strict) echo "--arith strict";; _c4) echo "--config c4";;
  _c4fast) echo "--config c4 --arith fast";; _c2) echo "--config c2";; _c2wgsl) echo "--config c2 --kernel wgsl";;
  _c5) echo "--config c5";; esac; }
SQ="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
# dynamic VALU mix by hardware class counter (tools/issue_floor.py prices it with the microbenchmark's costs)
CLS32="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64"
# flops as the hardware counts them, and the issue-occupancy pair of one and the same run
OCC="SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_FLOPS_FP32_TRANS SQ_INSTS_VALU_FLOPS_FP64_TRANS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 GRBM_GUI_ACTIVE"
CLS64="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU"
# memory-side instruction counts: the GLSL march fetches noise texels and spills 72 B per lane
MEM="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
# Order: every counter pass first, then profiles/traffic.json is rebuilt ON THE BOX from them
# (tools/summarize_profiles.py -> $OUT/summary, copied over the box's profiles/traffic.json), and only then
# the kernel traces: the bench line each trace record embeds therefore quotes the counter pass of this
# very session (its pass id is in the line), never a file that is overwritten later.
TAG=${PROFILE_TAG:-r05}
for sfx in base _strict _c4 _c4fast _c2 _c2wgsl; do
  want $sfx || continue
  extra=$(extra_of $sfx); tag=$sfx; [ $sfx == base ] && tag=""
  run pmc_fetch$tag --pmc FETCH_SIZE -- --steps 2 --warmup 1 $extra
  run pmc_write$tag --pmc WRITE_SIZE -- --steps 2 --warmup 1 $extra
  run pmc_sq$tag    --pmc $SQ -- --steps 2 --warmup 1 $extra
  run pmc_cls32$tag --pmc $CLS32 -- --steps 2 --warmup 1 $extra
  run pmc_cls64$tag --pmc $CLS64 -- --steps 2 --warmup 1 $extra
  run pmc_occ$tag --pmc $OCC -- --steps 2 --warmup 1 $extra
  case $sfx in _c2|_c2wgsl) run pmc_mem$tag --pmc $MEM -- --steps 2 --warmup 1 $extra;; esac
done
if want k16 || want base; then  # the base entry carries the K = 16 schedule's traffic: re-taken with it
  run pmc_fetch_k16 --pmc FETCH_SIZE -- --steps 2 --warmup 1 --segment-tries 16
  run pmc_write_k16 --pmc WRITE_SIZE -- --steps 2 --warmup 1 --segment-tries 16
fi
mkdir -p $OUT/summary
cp $R/profiles/traffic.json $OUT/summary/traffic.json 2>/dev/null
( cd $R && python tools/summarize_profiles.py $OUT $OUT/summary $TAG > $OUT/summarize_pmc.log 2>&1 )
cp $OUT/summary/traffic.json $R/profiles/traffic.json
for sfx in base _strict _c4 _c4fast _c2 _c2wgsl _c5; do
  want $sfx || continue
  extra=$(extra_of $sfx); tag=$sfx; [ $sfx == base ] && tag=""
  steps="--steps 5 --warmup 1"; [ $sfx == base ] && steps="--steps 10 --warmup 2"
  run trace$tag --kernel-trace --stats -- $steps $extra
done
( want k16 || want base ) && run trace_k16 --kernel-trace --stats -- --steps 10 --warmup 2 --segment-tries 16
( cd $R && python tools/summarize_profiles.py $OUT $OUT/summary $TAG > $OUT/summarize.log 2>&1 ); tail -3 $OUT/summarize.log
# the microbenchmark: costs in shader cycles (plain runs), and the same binary under the class counters
# (which counter does a mnemonic land in, what do SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES say per instruction)
if want microbench && hipcc -O2 --offload-arch=gfx950 $R/tools/valu_microbench.hip -o /tmp/valu_microbench 2> $OUT/valu_microbench_build.err; then
  for w in 1 2 4 8; do /tmp/valu_microbench $w > $OUT/valu_costs_w$w.json 2>> $OUT/valu_microbench.err; done
  mb() { local label=$1; shift
    timeout 300 rocprofv3 "$@" -d $OUT/$label -o bench -- /tmp/valu_microbench 4 > /dev/null 2> $OUT/$label.err
    db=$(find $OUT/$label -name "bench_results.db" | head -1); [ -n "$db" ] && [ "$db" != "$OUT/$label/bench_results.db" ] && mv "$db" $OUT/$label/bench_results.db; }
  mb mb_cls32 --pmc $CLS32
  mb mb_cls64 --pmc $CLS64
  mb mb_sq --pmc $SQ
  mb mb_occ --pmc $OCC
fi
find $OUT -name "*.csv" -delete
# the rocpd databases stay on the box (gpurun merges at most 64 MiB back): what travels is $OUT/summary
# (traffic.json, rNN_pmc_counters.json, rNN_trace_*_kernel_stats.txt), the bench lines and the .err files
find $OUT -name "*.db" | wc -l
find $OUT -name "*.db" -delete
find $OUT -mindepth 1 -type d -empty -delete
du -sh $OUT
