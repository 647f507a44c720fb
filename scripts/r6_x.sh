#!/bin/bash
# final driver-style bench line of the round, and which test skipped
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6x}
python3 bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench_stderr.txt; echo "bench rc=$?"
python3 - <<PY
import json
d = json.loads(open('gpurun_out/${TAG}_bench_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline'])
print({k: d['config'][k] for k in list(d['config'])[:21]})
print(d['type1']['stages_ms'], d['type2']['stages_ms'])
PY
python3 -m pytest tests -m gpu -q -p no:cacheprovider -rs -k "robustness or host_abi or fullsize" 2>&1 | grep -i "skip" | head -5
