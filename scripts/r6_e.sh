#!/bin/bash
# round 6: dense-set engine (first run) + the staged ring with 12 waves where 16 spilled
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r6e}
timeout 1200 python3 -m pytest tests -m gpu -q -x -p no:cacheprovider -k "dense_window or column_layer_sort_and_staged" > $O/${TAG}_tests.txt 2>&1
echo "pytest rc=$?" >> $O/${TAG}_tests.txt
tail -25 $O/${TAG}_tests.txt
P=$O/${TAG}_probes.txt; : > $P
probe() { echo "=== $*" >> $P; env "${ENVV[@]}" python3 scripts/perf_probe.py "$@" >> $P 2>&1; }
# dense engine A/B: the reference's protocol (sigma = 1.5, folded N(0, 1)), rho = 1 and 3.16; uniform rho = 1; m = 5, 6 at C2
for dense in 1 0; do
  ENVV=(NUFFT_DENSE=$dense)
  echo "##### NUFFT_DENSE=$dense" >> $P
  probe --z f64 --m 4 --sigma 1.5 --np 16777216 --dist randn --reps 4
  probe --z f64 --m 4 --sigma 1.5 --np 53054326 --dist randn --reps 3
  probe --z f64 --m 4 --sigma 1.5 --np 16777216 --reps 4
  probe --z f64 --m 4 --sigma 1.5 --np 5305433 --dist randn --reps 4
  probe --z c128 --m 4 --sigma 1.5 --np 16777216 --dist randn --reps 3
  probe --z f64 --m 5 --reps 4
  probe --z f64 --m 6 --reps 4
  probe --z f32 --m 6 --reps 4
  probe --z f64 --m 3 --np 4e7 --reps 3
done
for cs in 1 0; do
  ENVV=(NUFFT_COARSE_SORT=$cs)
  echo "##### NUFFT_COARSE_SORT=$cs" >> $P
  for cfg in "f32 2" "f32 3" "f64 3" "c64 3" "c64 4" "f64 2"; do
    set -- $cfg
    for mode in direct poly; do probe --z $1 --m $2 --mode $mode --reps 4; done
  done
done
grep -E "#####|===|set_points |t1_spread|t2_interp|engines|rror" $P
