#!/bin/bash
# experiments: (1) retire stores behind the second barrier of the spreading window (libnufft_defer.so), C2 both window modes, parity first;
# (2) ComplexF32 m = 8 interpolation ring with 12 waves and row groups of 8 (libnufft_m768.so), C3
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6t}; P=$O/${TAG}_probes.txt; : > $P
L=$R/nonuniformffts.jl_amd
NUFFT_LIB_PATH=$L/libnufft_defer.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "halo_variant_every or oversampled_grid or column_layer_sort_and" > $O/${TAG}_tests_defer.txt 2>&1; tail -3 $O/${TAG}_tests_defer.txt
for mode in direct poly; do for lib in mi355x defer mi355x defer; do
  echo "=== C2 $mode lib=$lib" >> $P
  NUFFT_LIB_PATH=$L/libnufft_$lib.so python3 scripts/perf_probe.py --z f64 --m 4 --np 1e7 --mode $mode --reps 8 2>&1 | grep -E "t1_spread|with set_points" | head -2 >> $P
done; done
NUFFT_LIB_PATH=$L/libnufft_m768.so timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "interpolation_ring_every" > $O/${TAG}_tests_m768.txt 2>&1; tail -3 $O/${TAG}_tests_m768.txt
for lib in mi355x m768; do
  echo "=== C3 poly lib=$lib" >> $P
  NUFFT_LIB_PATH=$L/libnufft_$lib.so timeout 600 python3 scripts/perf_probe.py --n 512 --np 1e8 --z c64 --m 8 --mode poly --reps 3 2>&1 | grep -E "t2_interp|type-2" >> $P
  echo "=== 256^3 c64 m=8 poly/direct lib=$lib" >> $P
  for mode in poly direct; do NUFFT_LIB_PATH=$L/libnufft_$lib.so python3 scripts/perf_probe.py --z c64 --m 8 --np 1e7 --mode $mode --reps 5 2>&1 | grep -E "t2_interp" >> $P; done
done
cat $P
