#!/bin/bash
# CPU-side sanitizer pass (GPU ASan is not available on this pool): builds the oracle -- the
# specified libm included, whose text the STRICT kernels share -- with UBSan and with ASan and runs
# the oracle-backed CPU tests against each build.  Usage: bash oracle/sanitize.sh
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
SRC="gravitas_oracle.c frame_oracle.c control_oracle.c shader_oracle.c viz_oracle.c post_oracle.c ref_libm.c wgsl_f64_twin.c"
TESTS="tests/test_oracle_pins.py tests/test_oracle_physics.py tests/test_golden_cpu.py tests/test_golden_shaders.py \
tests/test_post_chain.py tests/test_spacetime_viz.py tests/test_control_plane.py tests/test_shader_kernels.py tests/test_ref_libm.py tests/test_f32_oracle_pins.py"
cp "$R/oracle/libgravitas_oracle.so" /tmp/libgravitas_oracle.keep
restore() { cp /tmp/libgravitas_oracle.keep "$R/oracle/libgravitas_oracle.so"; touch "$R/oracle/libgravitas_oracle.so"; }
trap restore EXIT
for mode in undefined address; do
  extra=""; [ $mode = undefined ] && extra="-fno-sanitize-recover=undefined"
  (cd "$R/oracle" && gcc -O1 -g -std=c11 -fPIC -ffp-contract=off -fopenmp -fsanitize=$mode $extra -shared \
      -o libgravitas_oracle.so $SRC -lm && touch libgravitas_oracle.so)
  pre=""; [ $mode = address ] && pre="$(gcc -print-file-name=libasan.so)"
  echo "== -fsanitize=$mode"
  (cd "$R" && ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD=$pre \
      python -m pytest $TESTS -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -2)
done
