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
run pmc_fetch_k16 --pmc FETCH_SIZE -- --steps 2 --warmup 1 --segment-tries 16
run pmc_write_k16 --pmc WRITE_SIZE -- --steps 2 --warmup 1 --segment-tries 16
find $OUT -name "*.csv" -delete
find $OUT -name "*.db" | head -40
du -sh $OUT
