#!/bin/bash
# forward strided FFT passes: wave priority on the stores only (libnufft_ffs.so) / on the loads only (libnufft_ffl.so)
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6am}; P=$O/${TAG}_probes.txt; : > $P
L=$R/nonuniformffts.jl_amd
run() { for lib in mi355x ffs ffl mi355x ffs ffl; do echo "=== $1 lib=$lib" >> $P; NUFFT_LIB_PATH=$L/libnufft_$lib.so timeout 600 python3 scripts/perf_probe.py $2 2>&1 | grep -E "t1_fft|t1_deconv" | awk '{printf "%s %s  ", $1, $2}' >> $P; echo >> $P; done; }
run "C2 direct" "--z f64 --m 4 --np 1e7 --mode direct --reps 10"
run "C4 direct" "--z f64 --m 4 --np 1e7 --mode direct --c 3 --reps 6"
run "c128 256" "--z c128 --m 4 --np 1e7 --mode direct --reps 8"
run "f32 256" "--z f32 --m 4 --np 1e7 --mode direct --reps 8"
run "C3 poly" "--n 512 --np 1e8 --z c64 --m 8 --mode poly --reps 3"
cat $P
