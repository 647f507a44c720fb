#!/bin/bash
# spread_patch32_kernel with the travelling work between the matrix blocks (round 6) against round 5's loop (libnufft_p32old.so): parity, C3 A/B, phase table
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6u}; P=$O/${TAG}_probes.txt; : > $P
L=$R/nonuniformffts.jl_amd
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "patch_engine_every or both_spreading_engines or planar_patch" > $O/${TAG}_tests.txt 2>&1; tail -3 $O/${TAG}_tests.txt
for mode in poly direct; do for lib in mi355x p32old; do
  echo "=== C3 $mode lib=$lib" >> $P
  NUFFT_LIB_PATH=$L/libnufft_$lib.so timeout 600 python3 scripts/perf_probe.py --n 512 --np 1e8 --z c64 --m 8 --mode $mode --reps 3 2>&1 | grep -E "t1_spread|type-1" >> $P
done; done
echo "=== C3 poly, profile build (new loop)" >> $P
NUFFT_LIB_PATH=$L/libnufft_prof32.so timeout 600 python3 scripts/perf_probe.py --n 512 --np 1e8 --z c64 --m 8 --mode poly --reps 1 2>&1 | grep -E "patch" | tail -3 >> $P
for m in 7 10; do for lib in mi355x p32old; do
  echo "=== 256^3 c64 m=$m poly lib=$lib" >> $P
  NUFFT_LIB_PATH=$L/libnufft_$lib.so python3 scripts/perf_probe.py --z c64 --m $m --np 1e7 --mode poly --reps 5 2>&1 | grep -E "t1_spread|engines" >> $P
done; done
cat $P
