#!/bin/bash
# One GPU-box session (run through gpurun): the stages named on the command line, in order, every
# output under gpurun_out/<tag>/.   usage: tools/gpu_session.sh <tag> <stage> [<stage> ...]
#   smoke        __graft_entry__.smoke()
#   tests        the whole -m gpu suite (full-frame parity records written beside the log)
#   tests:<k>    pytest -m gpu -k <k>
#   bench        the bench.py line of every BASELINE config (+ the one-stream / wgsl / K = 16 secondaries)
#   bench:<args> one bench.py line with these arguments (quote them)
#   fuzz:<n>     tests/test_fuzz_parity.py with GRV_FUZZ_SEEDS=<n>
#   fuzzfast:<n> its FAST-contract tests only, figures of every test appended to fuzz_fast_report.jsonl
#   timeline     tools/march_timeline.py on ab_libs/tl_base.so and tl_lpt.so (built with -DGRV_MARCH_TIMELINE)
#   parity       the full-size parity records (c5, 8K, c4 every pixel)
#   profile[:<suffixes>]  tools/profile_gpu.sh (counter passes, then traces; see there)
#   renderers    tools/bench_renderers.py + tools/bench_shaders.py
#   identical    tools/ab_glsl_identical.py under every ab_libs/lib_*.so: three 1080p GLSL presets, every later library's
#                pixels and step counts compared bit for bit with the first one's
#   frames[:<args>]  napi/bench_frames.js for c3 and c2 (the frame loops driven from Node through the addon), each next to
#                bench.py's line of the same config on the same box
#   ab[:<configs>]  interleaved A/B of every ab_libs/lib_*.so (tools/ab_configs.sh; configs ';'-separated)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=${1:-session}; shift || true
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
for stage in "$@"; do
  name=${stage%%:*}; arg=""; [ "$stage" != "$name" ] && arg=${stage#*:}
  echo "== $stage"
  case $name in
    smoke) python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log;;
    tests)
      if [ -n "$arg" ]; then
        timeout 2700 python -m pytest tests -m gpu -q -x -k "$arg" > $O/pytest_k.log 2>&1; echo "pytest -k rc=$?" >> $O/pytest_k.log; tail -8 $O/pytest_k.log
      else
        GRV_C2_JSON=$O/full_frame_parity_c2.jsonl GRV_PARITY_JSON=$O/full_frame_parity.json timeout 3000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
        echo "pytest rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log
      fi;;
    bench)
      if [ -n "$arg" ]; then
        tag=$(echo $arg | tr -d ' -'); timeout 900 python bench.py $arg > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "rc=$?"; cut -c1-160 $O/bench_$tag.json
      else
        for cfg in "c3" "c2" "c2 --kernel wgsl" "c4" "c5 --steps 5 --warmup 1" "c2 --one-stream" "c2 --kernel wgsl --one-stream" \
                   "c3 --segment-tries 16 --no-cpu-baseline"; do
          tag=$(echo $cfg | tr -d ' -'); timeout 900 python bench.py --config $cfg > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "$cfg rc=$?"; cut -c1-130 $O/bench_$tag.json
        done
      fi;;
    fuzz) ( GRV_FUZZ_SEEDS=${arg:-100} GRV_FUZZ_REPORT=$O/fuzz_fast_report.jsonl timeout 3000 python -m pytest tests/test_fuzz_parity.py -m gpu -q --timeout 900 2>&1 | tail -25 ) > $O/fuzz.log 2>&1; cat $O/fuzz.log;;
    fuzzfast) ( GRV_FUZZ_SEED0=${FUZZ_SEED0:-0} GRV_FUZZ_SEEDS=${arg:-500} GRV_FUZZ_REPORT=$O/fuzz_fast_report.jsonl timeout 3000 python -m pytest tests/test_fuzz_parity.py -m gpu -q --timeout 900 -k "fast_stay or fast_hold" 2>&1 | tail -25 ) > $O/fuzzfast.log 2>&1; cat $O/fuzzfast.log;;
    timeline) for t in base lpt; do timeout 300 python tools/march_timeline.py ab_libs/tl_$t.so --frames 5 > $O/timeline_$t.json 2> $O/timeline_$t.err; echo "timeline $t rc=$?"; done;;
    parity)
      GRV_PARITY_TOL=1e-9 GRV_PARITY_JSON=$O/full_frame_parity_c5.json timeout 1200 python -m pytest tests/test_full_frame_parity.py -m gpu -q 2>&1 | tail -2
      GRV_PARITY_SIZE=7680x4320 GRV_PARITY_JSON=$O/full_frame_parity_8k.json timeout 2400 python -m pytest tests/test_full_frame_parity.py -m gpu -q 2>&1 | tail -2
      GRV_C4_STRIDE=1 GRV_C4_JSON=$O/full_frame_parity_c4.jsonl timeout 2400 python -m pytest tests/test_shader_kernels.py -m gpu -q -k test_config4_bench_form 2>&1 | tail -2;;
    profile) bash tools/profile_gpu.sh prof_$T $(echo $arg | tr ',' ' ') > $O/profile_gpu.log 2>&1; tail -3 $O/profile_gpu.log;;
    renderers)
      timeout 600 python tools/bench_renderers.py > $O/renderers.jsonl 2> $O/renderers.err; cut -c1-200 $O/renderers.jsonl
      timeout 600 python tools/bench_shaders.py > $O/shader_kernels.jsonl 2> $O/shaders.err; cut -c1-160 $O/shader_kernels.jsonl;;
    identical)
      cp blackhole-simulation_amd/libgravitas_hip.so /tmp/lib_keep.so; first=""
      for so in ab_libs/lib_*.so; do n=$(basename $so .so); cp $so blackhole-simulation_amd/libgravitas_hip.so
        timeout 300 python tools/ab_glsl_identical.py /tmp/ident_$n.npz 2> $O/identical_$n.err
        if [ -z "$first" ]; then first=$n; else echo "$n vs $first: $(python tools/ab_glsl_identical.py /tmp/ident_$first.npz /tmp/ident_$n.npz)" | tee -a $O/identical.txt; fi
      done; cp /tmp/lib_keep.so blackhole-simulation_amd/libgravitas_hip.so;;
    frames)
      for cfg in c3 c2 c4 c5; do
        sw=""; forms="--form all"
        [ $cfg = c5 ] && sw="--steps 5 --warmup 1"
        case $cfg in c4|c5) forms="--form device";; esac   # the two scaling / parity configs: the device-resident form only
        timeout 900 python bench.py --config $cfg $sw --no-cpu-baseline > $O/frames_benchpy_$cfg.json 2> $O/frames_benchpy_$cfg.err; echo "bench.py $cfg rc=$?"; cut -c1-120 $O/frames_benchpy_$cfg.json
        timeout 900 node napi/bench_frames.js --config $cfg $sw $forms $arg --out $O/napi_frames_$cfg.jsonl > $O/napi_frames_$cfg.log 2>&1; echo "bench_frames.js $cfg rc=$?"; cut -c1-150 $O/napi_frames_$cfg.log
      done;;
    ab) AB_CONFIGS="${arg:-c2;c2 --one-stream}" bash tools/ab_configs.sh $T/ab > $O/ab.log 2>&1; tail -40 $O/ab.log;;
    *) echo "unknown stage $stage";;
  esac
done
