#!/bin/bash
# dense engine: where does the time go?  Variants (threads, ablations) and SQ counters at uniform rho = 1 (sigma = 1.5, Np = 1.68e7), both windows;
# and the footprint test of the tree
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6h}; P=$O/${TAG}_probes.txt; : > $P
ARGS="--z f64 --m 4 --sigma 1.5 --np 16777216 --reps 4"
for lib in mi355x t768 abl1 abl2 abl3 abl4; do
  for mode in direct poly; do
    echo "=== lib=$lib mode=$mode" >> $P
    NUFFT_LIB_PATH=$R/nonuniformffts.jl_amd/libnufft_$lib.so python3 scripts/perf_probe.py $ARGS --mode $mode 2>&1 | grep -E "t1_spread|engines" >> $P
  done
done
echo "=== atomic window (NUFFT_DENSE=0)" >> $P
for mode in direct poly; do NUFFT_DENSE=0 python3 scripts/perf_probe.py $ARGS --mode $mode 2>&1 | grep -E "t1_spread|engines" >> $P; done
cat $P
scripts/pmc_probe.sh gpurun_out/${TAG}_pmc $ARGS --mode direct > $O/${TAG}_pmc.txt 2>&1
grep -A30 "spread_march_dense" $O/${TAG}_pmc.txt | head -80
timeout 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider -k "workspace_footprint or graph or c_abi" -s > $O/${TAG}_tests.txt 2>&1; grep -E "workspace|passed|failed|Error" $O/${TAG}_tests.txt | head -20
rm -rf $O/${TAG}_pmc/*/*/*.db 2>/dev/null; find $O/${TAG}_pmc -name "*.csv" -size +2M -delete
