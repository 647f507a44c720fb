#!/bin/bash
# usage (GPU box): scripts/r4_decision_sweep.sh -> spread + FFT stage (ms) with the LDS tiles forced, the window forced, and the plan's own
# choice (device-side decision per point set), for point distributions from uniform to tightly clustered
for dist in uniform randn cluster:1.5 cluster:1.0 cluster:0.7 cluster:0.5 cluster:0.3; do
  for sig in 2.0 1.5; do
    line="dist=$dist sigma=$sig :"
    for meth in 1 3 0; do
      out=$(NUFFT_SPREAD_METHOD=$meth python scripts/perf_probe.py --reps 3 --mode poly --sigma $sig --dist $dist "$@" 2>&1)
      sp=$(echo "$out" | grep -E "t1_spread" | awk '{print $2}')
      ff=$(echo "$out" | grep -E "t1_fft" | awk '{print $2}')
      eng=$(echo "$out" | grep -oE "engines: spread [a-z_]+" | awk '{print $3}')
      line="$line  [method=$meth $eng spread=${sp:-NA} fft=${ff:-NA}]"
    done
    echo "$line"
  done
done
