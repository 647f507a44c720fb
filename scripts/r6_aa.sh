#!/bin/bash
# a wave prepares its first chunk of the next layer before the retire pass (libnufft_ahead.so): parity (incl. non-uniform sets and segments), then C2 A/B in both window modes, m = 2..6 Direct
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6aa}; P=$O/${TAG}_probes.txt; : > $P
L=$R/nonuniformffts.jl_amd
NUFFT_LIB_PATH=$L/libnufft_ahead.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "halo_variant_every or oversampled_grid or column_layer_sort_and or spreading_ring_every or nonuniform or automatic or tasks_of_equal or clustered or type1_type2" > $O/${TAG}_tests.txt 2>&1; tail -3 $O/${TAG}_tests.txt
for mode in direct poly; do for lib in mi355x ahead mi355x ahead; do
  echo "=== C2 $mode lib=$lib" >> $P
  NUFFT_LIB_PATH=$L/libnufft_$lib.so python3 scripts/perf_probe.py --z f64 --m 4 --np 1e7 --mode $mode --reps 8 2>&1 | grep -E "t1_spread|with set_points" | head -2 >> $P
done; done
for m in 2 3 5 6; do for lib in mi355x ahead; do echo "=== 256^3 f64 m=$m direct lib=$lib" >> $P; NUFFT_LIB_PATH=$L/libnufft_$lib.so python3 scripts/perf_probe.py --z f64 --m $m --np 1e7 --mode direct --reps 6 2>&1 | grep -E "t1_spread" >> $P; done; done
for lib in mi355x ahead; do echo "=== refproto f64 randn lib=$lib" >> $P; NUFFT_LIB_PATH=$L/libnufft_$lib.so python3 scripts/perf_probe.py --z f64 --m 4 --sigma 1.5 --np 1e7 --dist randn --mode direct --reps 6 2>&1 | grep -E "t1_spread|with set_points" | head -2 >> $P; done
cat $P
