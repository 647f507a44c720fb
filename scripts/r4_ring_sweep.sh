#!/bin/bash
# usage (GPU box): scripts/r4_ring_sweep.sh  -> spread stage of the marching ring at C2 for several columns / segment counts
for cfg in "0 0 0" "32 32 1" "32 32 2" "32 32 4" "40 36 4" "40 36 8" "40 32 2" "40 32 4" "32 36 2" "36 36 4"; do
  set -- $cfg
  r=$(NUFFT_SPREAD_METHOD=3 NUFFT_SMARCH_N1=$1 NUFFT_SMARCH_N2=$2 NUFFT_SMARCH_NSEG=$3 python scripts/perf_probe.py --mode poly "${@:4}" 2>&1 | grep -E "ring_column|t1_spread" | sed 's/.*ring_column/ring_column/' | tr '\n' ' ')
  echo "n1=$1 n2=$2 nseg=$3 : $r"
done
