#!/bin/bash
# usage (GPU box): scripts/r4_engine_matrix.sh [extra perf_probe args]  -> spread stage (ms) of the three engines per (element type, M)
for z in f64 f32 c128 c64; do
  for m in 2 3 4 5 6 8; do
    line="$z m=$m :"
    for meth in 1 2 3; do
      t=$(NUFFT_SPREAD_METHOD=$meth python scripts/perf_probe.py --mode poly --z $z --m $m --reps 3 "$@" 2>&1 | grep -E "t1_spread" | awk '{print $2}')
      line="$line  method$meth=${t:-NA}"
    done
    echo "$line"
  done
done
