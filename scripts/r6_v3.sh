#!/bin/bash
# round 6, item 3: the column-layer sort + staged ring for every (T, M): the tests that cover them, then set_points / interpolation of the plans that gained it
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r6d}
python3 -m pytest tests -m gpu -q -p no:cacheprovider -k "column_layer or staged or interpolation_ring or slab_sort or graph or spreading_ring" > $O/${TAG}_tests.txt 2>&1
echo "pytest rc=$?" >> $O/${TAG}_tests.txt
tail -15 $O/${TAG}_tests.txt
for cfg in "f32 4" "f64 4" "c128 4" "c64 4" "f64 3" "f64 2" "f32 3" "f32 2" "c64 3"; do
  set -- $cfg
  for mode in direct poly; do
    echo "=== z=$1 m=$2 mode=$mode" >> $O/${TAG}_probes.txt
    python3 scripts/perf_probe.py --z $1 --m $2 --mode $mode --reps 5 >> $O/${TAG}_probes.txt 2>&1
    echo "--- NUFFT_COARSE_SORT=0" >> $O/${TAG}_probes.txt
    NUFFT_COARSE_SORT=0 python3 scripts/perf_probe.py --z $1 --m $2 --mode $mode --reps 5 >> $O/${TAG}_probes.txt 2>&1
  done
done
grep -E "===|---|set_points |t2_interp|engines" $O/${TAG}_probes.txt
