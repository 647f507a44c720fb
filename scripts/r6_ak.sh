#!/bin/bash
# wave priority in the contiguous-line FFT kernel of complex plans (cplx_lines_kernel; libnufft_fpc.so): ComplexF64 256^3, C3, ComplexF32 256^3
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6ak}; P=$O/${TAG}_probes.txt; : > $P
L=$R/nonuniformffts.jl_amd
run() { for lib in mi355x fpc mi355x fpc; do echo "=== $1 lib=$lib" >> $P; NUFFT_LIB_PATH=$L/libnufft_$lib.so timeout 600 python3 scripts/perf_probe.py $2 2>&1 | grep -E "t1_fft|t1_deconv|t2_deconv_pad|t2_fft" | awk '{printf "%s %s  ", $1, $2}' >> $P; echo >> $P; done; }
NUFFT_LIB_PATH=$L/libnufft_fpc.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "type1_type2 or fftshift or oversampled_grid" > $O/${TAG}_tests.txt 2>&1; tail -2 $O/${TAG}_tests.txt
run "c128 256" "--z c128 --m 4 --np 1e7 --mode direct --reps 8"
run "C3 poly" "--n 512 --np 1e8 --z c64 --m 8 --mode poly --reps 3"
run "c64 256" "--z c64 --m 4 --np 1e7 --mode direct --reps 8"
cat $P
