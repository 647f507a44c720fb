#!/bin/bash
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6m}
timeout 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider -k "dense_point_sets or dense_window" > $O/${TAG}_tests.txt 2>&1; tail -4 $O/${TAG}_tests.txt
# counters A/B at rho = 1 uniform (sigma = 1.5, Np = 1.68e7), Direct(): the dense engine and the atomic window
NUFFT_DENSE_MIN=0 scripts/pmc_probe.sh gpurun_out/${TAG}_pmc_dense --z f64 --m 4 --sigma 1.5 --np 16777216 --mode direct > $O/${TAG}_pmc_dense.txt 2>&1
NUFFT_DENSE=0 scripts/pmc_probe.sh gpurun_out/${TAG}_pmc_atomic --z f64 --m 4 --sigma 1.5 --np 16777216 --mode direct > $O/${TAG}_pmc_atomic.txt 2>&1
NUFFT_DENSE_MIN=0 scripts/pmc_probe.sh gpurun_out/${TAG}_pmc_dense_poly --z f64 --m 4 --sigma 1.5 --np 53054326 --mode poly > $O/${TAG}_pmc_dense_poly.txt 2>&1
NUFFT_DENSE=0 scripts/pmc_probe.sh gpurun_out/${TAG}_pmc_atomic_poly --z f64 --m 4 --sigma 1.5 --np 53054326 --mode poly > $O/${TAG}_pmc_atomic_poly.txt 2>&1
find $O -path "*${TAG}_pmc*" \( -name "*.db" -o -name "*.csv" \) -delete
grep -A9 "spread_march" $O/${TAG}_pmc_dense.txt | head -40
