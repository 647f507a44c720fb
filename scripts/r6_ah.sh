#!/bin/bash
# wave priority in the FFT passes (waves that load / store ahead of waves that transform; s_setprio 1 / 3 against none): C2, C3, C4, Float32
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6ah}; P=$O/${TAG}_probes.txt; : > $P
L=$R/nonuniformffts.jl_amd
run() { for lib in mi355x fpa fpb; do echo "=== $1 lib=$lib" >> $P; NUFFT_LIB_PATH=$L/libnufft_$lib.so timeout 600 python3 scripts/perf_probe.py $2 2>&1 | grep -E "t1_fft|t1_deconv|t2_deconv_pad|t2_fft" | awk '{printf "%s %s  ", $1, $2}' >> $P; echo >> $P; done; }
NUFFT_LIB_PATH=$L/libnufft_fpa.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "type1_type2 or fftshift or oversampled_grid" > $O/${TAG}_tests.txt 2>&1; tail -2 $O/${TAG}_tests.txt
run "C2 direct" "--z f64 --m 4 --np 1e7 --mode direct --reps 10"
run "C2 direct again" "--z f64 --m 4 --np 1e7 --mode direct --reps 10"
run "C4 direct" "--z f64 --m 4 --np 1e7 --mode direct --c 3 --reps 6"
run "f64 m=4 n=128" "--n 128 --z f64 --m 4 --np 1e6 --mode direct --reps 8"
run "f32 256" "--z f32 --m 4 --np 1e7 --mode direct --reps 8"
run "c128 256" "--z c128 --m 4 --np 1e7 --mode direct --reps 8"
cat $P
