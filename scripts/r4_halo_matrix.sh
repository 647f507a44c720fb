#!/bin/bash
# usage (GPU box): scripts/r4_halo_matrix.sh  -> spread and FFT stages (ms) of the marching ring with clipped columns (halo 0) and the
# halo variant (halo 2: side buffer added by the first FFT pass), per (real element type, M, window, components)
run() {  # label, perf_probe args
  local label=$1; shift
  line="$label :"
  for h in 0 2; do
    out=$(NUFFT_SPREAD_METHOD=3 NUFFT_SMARCH_HALO=$h python scripts/perf_probe.py --reps 3 "$@" 2>&1)
    sp=$(echo "$out" | grep -E "t1_spread" | awk '{print $2}')
    ff=$(echo "$out" | grep -E "t1_fft" | awk '{print $2}')
    dc=$(echo "$out" | grep -E "t1_deconv" | awk '{print $2}')
    hh=$(echo "$out" | grep -oE "halo=[0-9]" | head -1)
    col=$(echo "$out" | grep -oE "ring_column=\[[0-9, ]+\] x[0-9]+" | head -1)
    line="$line  [$hh $col spread=${sp:-NA} fft=${ff:-NA} deconv=${dc:-NA}]"
  done
  echo "$line"
}
for z in f64 f32; do
  for m in 2 3 4 5 6 7 8; do run "$z m=$m poly" --mode poly --z $z --m $m; done
done
run "f64 m=4 direct" --mode direct --z f64 --m 4
run "f64 m=4 C=2" --mode poly --z f64 --m 4 --c 2
run "f64 m=4 C=3" --mode poly --z f64 --m 4 --c 3
run "f64 m=4 sigma=1.5 Np=1.68e7 randn" --mode direct --z f64 --m 4 --sigma 1.5 --np 16777216 --dist randn
run "f64 m=4 sigma=1.5 Np=1.68e7 uniform" --mode direct --z f64 --m 4 --sigma 1.5 --np 16777216
run "f64 m=4 n=128" --mode poly --z f64 --m 4 --n 128 --np 1e6
run "f64 m=4 n=384" --mode poly --z f64 --m 4 --n 384 --np 3e7
