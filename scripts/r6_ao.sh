#!/bin/bash
# dense-set engine: wave priority while a wave issues matrix instructions / flushes a bin (libnufft_dprio.so)
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6ao}; P=$O/${TAG}_probes.txt; : > $P
L=$R/nonuniformffts.jl_amd
run() { for lib in mi355x dprio mi355x dprio; do echo "=== $1 lib=$lib" >> $P; NUFFT_LIB_PATH=$L/libnufft_$lib.so timeout 600 python3 scripts/perf_probe.py $2 2>&1 | grep -E "t1_spread|engines" | awk '{printf "%s %s %s  ", $1, $2, $3}' >> $P; echo >> $P; done; }
NUFFT_LIB_PATH=$L/libnufft_dprio.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "dense_window_engine" > $O/${TAG}_tests.txt 2>&1; tail -2 $O/${TAG}_tests.txt
run "f64 m=6 direct 4e7" "--z f64 --m 6 --np 4e7 --mode direct --reps 5"
run "f64 m=5 direct 4e7" "--z f64 --m 5 --np 4e7 --mode direct --reps 5"
run "f64 m=4 poly 6e7" "--z f64 --m 4 --np 6e7 --mode poly --reps 5"
run "f64 m=4 direct sigma1.5 rho10" "--z f64 --m 4 --sigma 1.5 --np 1.6e8 --mode direct --reps 3"
cat $P
