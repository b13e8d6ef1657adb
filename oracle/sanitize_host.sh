#!/bin/bash
# ASan + UBSan pass over the PRODUCT's host code on the CPU box (GPU ASan is not available on this
# pool): the host side of libgravitas_hip.so -- every engine*.hip entry point (engine_multi.hip included), control_plane.hip's
# host twins, grv_strict_math_host, the tile (un)pack helpers -- and napi/gravitas_napi.c; then the multi-GPU host logic
# (csrc/multi_core.hpp) under ThreadSanitizer over a mock stream / transport Api (tests/host/multi_tsan.cpp).
#   device code : compiled for gfx950 as shipped (-Xarch_host keeps the sanitizers off it)
#   host code   : -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined
# The sanitized library takes the in-tree library's place for the run (restored on exit); the tests
# are the no-device ones: host logic, the C-ABI surface, control plane / viz host twins, the gloo
# world-size-2 assembly, and the CPU half of the N-API tests (node loads the sanitized addon with the
# ASan runtime preloaded).  Usage: bash oracle/sanitize_host.sh
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
CS="$R/blackhole-simulation_amd/csrc"
LIB="$R/blackhole-simulation_amd/libgravitas_hip.so"
ADDON="$R/napi/blackhole_physics.node"
B=/tmp/grv_host_san
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
CLANG=/opt/rocm/lib/llvm/bin/clang
RT=$($CLANG -print-file-name=libclang_rt.asan-x86_64.so)
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
XSAN=""; for f in $SAN; do XSAN="$XSAN -Xarch_host $f"; done
TESTS=${SAN_TESTS:-"tests/test_host_logic.py tests/test_control_plane.py tests/test_spacetime_viz.py tests/test_ref_libm.py \
tests/test_dist_cpu.py tests/test_napi_addon.py tests/test_closed_form_properties.py"}

mkdir -p $B
cp "$LIB" $B/libgravitas_hip.keep
[ -f "$ADDON" ] && cp "$ADDON" $B/addon.keep
restore() {
  cp $B/libgravitas_hip.keep "$LIB"; touch "$LIB"
  [ -f $B/addon.keep ] && { cp $B/addon.keep "$ADDON"; touch "$ADDON"; }
  return 0
}
trap restore EXIT

for tu in kernels_strict kernels_fast kernels_fast_f64 control_plane spacetime_viz engine engine_shaders engine_control engine_images engine_multi; do
  fp="-ffp-contract=off"
  [ $tu = kernels_fast ] && fp="-ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt"
  [ $tu = kernels_fast_f64 ] && fp="-ffp-contract=fast"
  (cd "$CS" && $HIPCC -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $fp $XSAN \
      -I/opt/rocm/include -c $tu.hip -o $B/$tu.o) &
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC $SAN -shared-libsan -o "$LIB" $B/kernels_strict.o $B/kernels_fast.o $B/kernels_fast_f64.o \
    $B/control_plane.o $B/spacetime_viz.o $B/engine.o $B/engine_shaders.o $B/engine_control.o $B/engine_images.o $B/engine_multi.o -ldl -lpthread
# -asan-globals=0 on the addon only: its merged string literals land on odd addresses, which ASan's
# global registration refuses under node; stack and heap checking (the argument buffers, the arena)
# stay on
if [ -f /usr/include/node/node_api.h ]; then
  $CLANG -O1 -g -std=c11 -fPIC -shared -Wall -Wextra $SAN -mllvm -asan-globals=0 -shared-libsan -I/usr/include/node \
      "$R/napi/gravitas_napi.c" -o "$ADDON" -L"$R/blackhole-simulation_amd" -lgravitas_hip \
      -Wl,-rpath,'$ORIGIN/../blackhole-simulation_amd'
fi
# ---- ThreadSanitizer leg: the multi-GPU host logic (csrc/multi_core.hpp: rank threads, per-parity slots and events,
# the frame skeleton) over a mock Api whose streams are threads and whose copies touch memory -- 1000 frames x
# G = 2, 4, 8 x both transports x both exchange formats with frames of both parities in flight and injected faults;
# the two mutants (an event wait removed) must be caught.  SAN_TSAN_FRAMES overrides the frame count.
echo "== multi-GPU host logic under ThreadSanitizer (tests/host/multi_tsan.cpp)"
TS="$R/tests/host/multi_tsan.cpp"
$CLANG++ -O1 -g -std=c++17 -pthread -fsanitize=thread "$TS" -o $B/multi_tsan
TSAN_OPTIONS=halt_on_error=1:second_deadlock_stack=1 $B/multi_tsan ${SAN_TSAN_FRAMES:-1000} | tail -1
for mut in GRVMULTI_MUTANT_NO_SLOT_WAIT GRVMULTI_MUTANT_NO_ARRIVED_WAIT; do
  $CLANG++ -O1 -g -std=c++17 -pthread -fsanitize=thread -D$mut "$TS" -o $B/multi_tsan_mut
  if TSAN_OPTIONS=halt_on_error=1 $B/multi_tsan_mut 60 > $B/mut.log 2>&1; then echo "mutant $mut went UNNOTICED"; exit 1
  else echo "mutant $mut: caught ($(grep -c 'ThreadSanitizer: data race' $B/mut.log) race report(s))"; fi
done
echo "== host code of libgravitas_hip.so + N-API addon under ASan + UBSan"
cd "$R"
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
  LD_PRELOAD=$RT python -m pytest $TESTS -x -q -m "not gpu" -p no:cacheprovider --deselect tests/test_host_logic.py::test_committed_counter_passes_are_of_the_library_in_the_tree 2>&1 | tail -${SAN_TAIL:-3}
