#!/bin/bash
# LDS-only barriers in the z-marching kernels (no wait for global store acknowledgements per layer): parity + timing
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6n}; P=$O/${TAG}_probes.txt; : > $P
timeout 1200 python3 -m pytest tests -m gpu -q -x -p no:cacheprovider -k "ring or column_layer or staged or dense_window or graph or c2_full or config_c4" > $O/${TAG}_tests.txt 2>&1; tail -4 $O/${TAG}_tests.txt
for cfg in "--z f64 --m 4" "--z f64 --m 4 --c 3" "--z c128 --m 4" "--z f32 --m 4" "--z f64 --m 6" "--z c64 --m 8 --n 512 --np 1e8" "--z f64 --m 4 --sigma 1.5 --np 16777216 --dist randn"; do for mode in direct poly; do
  echo "=== $cfg $mode" >> $P
  python3 scripts/perf_probe.py --reps 5 --mode $mode $cfg 2>&1 | grep -E "set_points |t1_spread|t1_fft|t2_interp|with set_points" >> $P
done; done
cat $P
