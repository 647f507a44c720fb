#!/bin/bash
# dense engine, software-pipelined batches: parity tests, then A/B against the atomic window over densities and half-supports, ablations
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6i}; P=$O/${TAG}_probes.txt; : > $P
timeout 900 python3 -m pytest tests -m gpu -q -x -p no:cacheprovider -k "dense_window" > $O/${TAG}_tests.txt 2>&1; tail -3 $O/${TAG}_tests.txt
run() { echo "=== $*" >> $P; env "${ENVV[@]}" python3 scripts/perf_probe.py --z f64 --reps 4 "$@" 2>&1 | grep -E "t1_spread|set_points |engines" >> $P; }
for mode in direct poly; do
  for cfg in "--m 4 --sigma 1.5 --np 16777216" "--m 4 --sigma 1.5 --np 5305433" "--m 4 --sigma 1.5 --np 53054326" "--m 4 --sigma 1.5 --np 16777216 --dist randn" "--m 4 --sigma 1.5 --np 53054326 --dist randn" "--m 4" "--m 5" "--m 6" "--m 3 --np 4e7" "--m 5 --np 4e7" "--m 6 --np 4e7"; do
    ENVV=(NUFFT_DENSE_MIN=0); run --mode $mode $cfg
    ENVV=(NUFFT_DENSE=0); run --mode $mode $cfg
  done
done
for lib in abl1 abl2 abl3; do for mode in direct poly; do
  echo "=== lib=$lib mode=$mode rho=1 uniform" >> $P
  NUFFT_DENSE_MIN=0 NUFFT_LIB_PATH=$R/nonuniformffts.jl_amd/libnufft_$lib.so python3 scripts/perf_probe.py --z f64 --m 4 --sigma 1.5 --np 16777216 --reps 4 --mode $mode 2>&1 | grep -E "t1_spread" >> $P
done; done
cat $P
