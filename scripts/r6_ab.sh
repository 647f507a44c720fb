#!/bin/bash
# final tree (spreading window restructured into prep + accum, experiments off): whole GPU suite, smoke, C-ABI driver, bench line
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6ab}
bash scripts/r6_gputests.sh $TAG
python3 bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench_stderr.txt; echo "bench rc=$?"
python3 - <<PY
import json
d = json.loads(open('gpurun_out/${TAG}_bench_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('profile_matches_run'))
print({k: d['config'][k] for k in list(d['config'])[1:21]})
print(d['type1']['stages_ms'], d['type2']['stages_ms'])
PY
