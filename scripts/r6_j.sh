#!/bin/bash
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6j}; P=$O/${TAG}_probes.txt; : > $P
ARGS="--z f64 --m 4 --sigma 1.5 --np 16777216 --reps 4"
for lib in mi355x a2 a18 a58 a62 a63; do for mode in direct poly; do
  echo "=== lib=$lib mode=$mode" >> $P
  NUFFT_DENSE_MIN=0 NUFFT_LIB_PATH=$R/nonuniformffts.jl_amd/libnufft_$lib.so python3 scripts/perf_probe.py $ARGS --mode $mode 2>&1 | grep -E "t1_spread" >> $P
done; done
for cfg in "--m 4 --sigma 1.5 --np 53054326" "--m 5" "--m 5 --np 4e7" "--m 6" "--m 4 --sigma 1.5 --np 16777216 --dist randn"; do for mode in direct poly; do
  echo "=== dense $cfg $mode" >> $P
  NUFFT_DENSE_MIN=0 python3 scripts/perf_probe.py --z f64 --reps 3 --mode $mode $cfg 2>&1 | grep -E "t1_spread" >> $P
done; done
cat $P
