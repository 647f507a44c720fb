#!/bin/bash
# wave priority during the accumulation of a chunk of the spreading window: s_setprio 1 / 2 / 3 against none, over element types, half-supports and point sets
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6ae}; P=$O/${TAG}_probes.txt; : > $P
L=$R/nonuniformffts.jl_amd
run() { for lib in mi355x prio1 prio2 prio3; do echo "=== $1 lib=$lib" >> $P; NUFFT_LIB_PATH=$L/libnufft_$lib.so python3 scripts/perf_probe.py $2 --reps 8 2>&1 | grep -E "t1_spread" >> $P; done; }
run "C2 direct" "--z f64 --m 4 --np 1e7 --mode direct"
run "C2 poly" "--z f64 --m 4 --np 1e7 --mode poly"
run "C2 direct again" "--z f64 --m 4 --np 1e7 --mode direct"
run "C4 direct" "--z f64 --m 4 --np 1e7 --mode direct --c 3"
run "f64 m=2 direct" "--z f64 --m 2 --np 1e7 --mode direct"
run "f64 m=3 direct" "--z f64 --m 3 --np 1e7 --mode direct"
run "f64 m=5 direct" "--z f64 --m 5 --np 1e7 --mode direct"
run "f64 m=6 direct" "--z f64 --m 6 --np 1e7 --mode direct"
run "f32 m=4 direct" "--z f32 --m 4 --np 1e7 --mode direct"
run "c128 m=4 direct" "--z c128 --m 4 --np 1e7 --mode direct"
run "c64 m=4 direct" "--z c64 --m 4 --np 1e7 --mode direct"
run "refproto f64 randn" "--z f64 --m 4 --sigma 1.5 --np 1e7 --dist randn --mode direct"
run "f64 m=4 np=1e6" "--z f64 --m 4 --np 1e6 --mode direct"
cat $P | paste - - | awk '{print $2,$3,$4,$6,$7,$8,$9}' 
