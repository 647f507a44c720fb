#!/bin/bash
# robustness sweeps of the round-6 machinery: the whole GPU suite with (a) the dense-set engine forced wherever it exists, (b) the slab sort preferred
# (no column-layer sort), (c) no adaptive sort choice.  Tests that assert an engine / sort NAME are expected to fail under a forced switch; parity failures are not.
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6y}
for v in "NUFFT_DENSE_MIN=1" "NUFFT_COARSE_SORT=0" "NUFFT_SORT_ADAPTIVE=0"; do
  n=$(echo $v | tr '=' '_')
  env $v timeout 1500 python3 -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_fullsize.py -k "not robustness_footprint" --maxfail=25 > $O/${TAG}_$n.txt 2>&1
  echo "== $v"; tail -30 $O/${TAG}_$n.txt | grep -E "passed|failed|FAILED" | head -30
done
