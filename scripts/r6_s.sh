#!/bin/bash
# C3: phase table of spread_patch32_kernel at today's code (NUFFT_PATCH_PROFILE build), both window modes; and the regular build beside it
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6s}; P=$O/${TAG}_probes.txt; : > $P
for mode in poly direct; do
  echo "=== C3 $mode, profile build" >> $P
  NUFFT_LIB_PATH=$R/nonuniformffts.jl_amd/libnufft_prof32.so timeout 600 python3 scripts/perf_probe.py --n 512 --np 1e8 --z c64 --m 8 --mode $mode --reps 2 2>&1 | grep -E "patch|set_points |t1_spread|t2_interp|with set_points|engines" | tail -12 >> $P
  echo "=== C3 $mode, regular build" >> $P
  timeout 600 python3 scripts/perf_probe.py --n 512 --np 1e8 --z c64 --m 8 --mode $mode --reps 3 2>&1 | grep -E "set_points |t1_spread|t2_interp|with set_points|engines" >> $P
done
cat $P
