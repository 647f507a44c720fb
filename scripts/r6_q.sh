#!/bin/bash
# adaptive sort choice (rings' decisions fed back to the next set_points): tests that touch the sorts, then non-uniform sets, then the bench line
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6q}; P=$O/${TAG}_probes.txt; : > $P
timeout 1500 python3 -m pytest tests -m gpu -q -p no:cacheprovider -k "sort or column_layer or graph or robustness or slab or nonuniform or clustered or consistency" > $O/${TAG}_tests.txt 2>&1; tail -4 $O/${TAG}_tests.txt
for cfg in "--sigma 1.5 --np 1e7 --dist randn" "--sigma 1.5 --np 16777216 --dist randn" "--sigma 1.5 --np 1678 --dist randn" "--np 1e7" "--np 1e7 --dist cluster:0.5"; do for ad in 1 0; do
  echo "=== $cfg NUFFT_SORT_ADAPTIVE=$ad" >> $P
  NUFFT_SORT_ADAPTIVE=$ad python3 scripts/perf_probe.py --z f64 --m 4 --reps 6 $cfg 2>&1 | grep -E "set_points |t1_spread|t2_interp|with set_points|engines" >> $P
done; done
cat $P
python3 bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench_stderr.txt
python3 - <<PY
import json
d = json.loads(open('gpurun_out/${TAG}_bench_line.json').read().strip().splitlines()[-1])
print({k: d['config'][k] for k in list(d['config'])[:20]})
print(d['type1']['stages_ms'], d['type2']['stages_ms'])
PY
