#!/bin/bash
# wave priority during the gather of a pass of the interpolation ring (s_setprio 1 / 3 against none): C2 both window modes, C3, other half-supports
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6af}; P=$O/${TAG}_probes.txt; : > $P
L=$R/nonuniformffts.jl_amd
run() { for lib in mi355x mprio1 mprio3; do echo "=== $1 lib=$lib" >> $P; NUFFT_LIB_PATH=$L/libnufft_$lib.so timeout 600 python3 scripts/perf_probe.py $2 2>&1 | grep -E "t2_interp" >> $P; done; }
NUFFT_LIB_PATH=$L/libnufft_mprio3.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "interpolation_ring_every or column_layer_sort_and" > $O/${TAG}_tests.txt 2>&1; tail -2 $O/${TAG}_tests.txt
run "C2 direct" "--z f64 --m 4 --np 1e7 --mode direct --reps 8"
run "C2 poly" "--z f64 --m 4 --np 1e7 --mode poly --reps 8"
run "C2 direct again" "--z f64 --m 4 --np 1e7 --mode direct --reps 8"
run "C3 poly" "--n 512 --np 1e8 --z c64 --m 8 --mode poly --reps 3"
run "f64 m=6 direct" "--z f64 --m 6 --np 1e7 --mode direct --reps 6"
run "f64 m=2 direct" "--z f64 --m 2 --np 1e7 --mode direct --reps 6"
run "refproto f64 randn" "--z f64 --m 4 --sigma 1.5 --np 1e7 --dist randn --mode direct --reps 6"
paste - - < $P | sed 's/t2_interp *//; s/=== //'
