#!/bin/bash
# usage (GPU box): scripts/r4_interp_threads.sh <libA.so> <libB.so>  -> interpolation stage (ms) of the ring per (element type, M, window) with two builds
for z in f64 f32 c128 c64; do
  for m in 2 3 4 5 6 8; do
    for mode in poly direct; do
      line="$z m=$m $mode :"
      for lib in "$@"; do
        t=$(NUFFT_LIB_PATH=$lib NUFFT_INTERP_MARCH=2 python scripts/perf_probe.py --mode $mode --z $z --m $m --reps 3 2>&1 | grep -E "t2_interp" | awk '{print $2}')
        line="$line  ${t:-NA}"
      done
      echo "$line"
    done
  done
done
