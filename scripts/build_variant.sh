#!/bin/bash
# usage: scripts/build_variant.sh <name> "<extra flags>" unit [unit ...]
# Builds nonuniformffts.jl_amd/libnufft_<name>.so: the objects of the regular build with the listed translation units
# (e.g. smarch_f64r) recompiled with the extra flags.  Use with NUFFT_LIB_PATH=... (A/B and ablation runs).
set -e
NAME=$1; FLAGS=$2; shift 2
cd "$(dirname "$0")/../nonuniformffts.jl_amd/csrc"
B=build_$NAME
rm -rf $B; mkdir -p $B
cp build/*.o $B/
for u in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I. -I../../include -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-variable $FLAGS -x hip -c $u.hip -o $B/$u.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 $B/*.o -shared -L/opt/rocm/lib -lrocfft -lamdhip64 -Wl,-rpath,/opt/rocm/lib -o ../libnufft_$NAME.so
echo built ../libnufft_$NAME.so
