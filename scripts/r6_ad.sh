#!/bin/bash
# wave priority 3 during the accumulation of a chunk (libnufft_prio3.so): parity, then C2 A/B in both window modes
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6ad}; P=$O/${TAG}_probes.txt; : > $P
L=$R/nonuniformffts.jl_amd
NUFFT_LIB_PATH=$L/libnufft_prio3.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "halo_variant_every or oversampled_grid or column_layer_sort_and or spreading_ring_every" > $O/${TAG}_tests.txt 2>&1; tail -3 $O/${TAG}_tests.txt
for mode in direct poly; do for lib in mi355x prio3 mi355x prio3; do
  echo "=== C2 $mode lib=$lib" >> $P
  NUFFT_LIB_PATH=$L/libnufft_$lib.so python3 scripts/perf_probe.py --z f64 --m 4 --np 1e7 --mode $mode --reps 8 2>&1 | grep -E "t1_spread|with set_points" | head -2 >> $P
done; done
cat $P
