#!/bin/bash
# A/B builds of the FAST translation unit: tools/ab_build.sh NAME "-DSWITCH=0 ..." compiles
# kernels_fast.hip with the extra flags and links ab_libs/lib_NAME.so from it and the other
# (unchanged) objects of csrc/build.  TU=strict does the same with kernels_strict.hip (-ffp-contract=off), TU=fast64 with
# kernels_fast_f64.hip (the Makefile's flags incl. -mllvm -enable-post-misched=0; override by appending), TU=engine with csrc/engine.hip (host switches).  ab_libs/ is git-ignored but travels to the GPU box, where
# tools/ab_bench.sh swaps each library in under bench.py in turn on ONE box (box-to-box spread of the
# f64 headline is +-4 %: never compare numbers of two boxes).
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/blackhole-simulation_amd/csrc
NAME=$1; shift
EXTRA="${1:-}"
mkdir -p $R/ab_libs
make -C $C -s
FAST_O=$C/build/kernels_fast.o
FAST64_O=$C/build/kernels_fast_f64.o
STRICT_O=$C/build/kernels_strict.o
ENGINE_O=$C/build/engine.o
if [ "${TU:-fast}" = engine ]; then   # host-side switch (csrc/engine.hip)
  ENGINE_O=$C/build/engine_$NAME.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off \
    $EXTRA -c $C/engine.hip -o $ENGINE_O
elif [ "${TU:-fast}" = fast64 ]; then
  FAST64_O=$C/build/kernels_fast_f64_$NAME.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize \
    -mllvm -enable-post-misched=0 $EXTRA -c $C/kernels_fast_f64.hip -o $FAST64_O
elif [ "${TU:-fast}" = strict ]; then
  STRICT_O=$C/build/kernels_strict_$NAME.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -fno-slp-vectorize \
    $EXTRA -c $C/kernels_strict.hip -o $STRICT_O
else
  FAST_O=$C/build/kernels_fast_$NAME.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast \
    -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize $EXTRA -c $C/kernels_fast.hip -o $FAST_O
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab_libs/lib_$NAME.so $STRICT_O \
  $FAST_O $FAST64_O $C/build/control_plane.o $C/build/spacetime_viz.o $ENGINE_O \
  $C/build/engine_shaders.o $C/build/engine_control.o $C/build/engine_images.o $C/build/engine_multi.o -ldl -lpthread
echo "ab_libs/lib_$NAME.so  ($EXTRA)"
