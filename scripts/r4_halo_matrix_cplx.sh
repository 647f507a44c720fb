#!/bin/bash
# complex plans: marching ring with clipped columns / halo variant, and the patches (spread + FFT stages, ms)
run() {
  local label=$1; shift
  line="$label :"
  for cfg in "3 0" "3 2" "2 0"; do
    set -- $cfg "${@:1}"
    meth=$1; h=$2; shift 2
    out=$(NUFFT_SPREAD_METHOD=$meth NUFFT_SMARCH_HALO=$h python scripts/perf_probe.py --reps 3 "$@" 2>&1)
    sp=$(echo "$out" | grep -E "t1_spread" | awk '{print $2}')
    ff=$(echo "$out" | grep -E "t1_fft" | awk '{print $2}')
    hh=$(echo "$out" | grep -oE "halo=[0-9]" | head -1)
    col=$(echo "$out" | grep -oE "ring_column=\[[0-9, ]+\] x[0-9]+" | head -1)
    line="$line  [method=$meth $hh $col spread=${sp:-NA} fft=${ff:-NA}]"
  done
  echo "$line"
}
for z in c128 c64; do
  for m in 2 3 4 5 6; do run "$z m=$m poly" --mode poly --z $z --m $m; done
done
