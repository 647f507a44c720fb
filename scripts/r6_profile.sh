#!/bin/bash
# round 6 profiles: the committed bench line (C2 Direct headline + compact C3 / C4 records + reference protocol), then rocprofv3 kernel stats, PMC traffic and
# SQ counters of C2 (Direct and polynomial windows separately), C3 and C4 (Direct).  usage (through gpurun): bash scripts/r6_profile.sh <letter>
L=${1:-a}
cd "$GRAFT_REPO_ROOT" || exit 1
bash scripts/profile_bench.sh round6_${L}_bench_c2_direct "C2, Direct window (the ROC default, bench.py's headline; round 6 $L)" > /dev/null 2>&1
bash scripts/profile_bench.sh round6_${L}_bench_c2 "C2, FastApproximation window (round 6 $L)" --evalmode fast --only-headline > /dev/null 2>&1
bash scripts/profile_bench.sh round6_${L}_bench_c4_direct "C4, Direct window (round 6 $L)" --config c4 --only-headline > /dev/null 2>&1
bash scripts/profile_bench.sh round6_${L}_bench_c3_direct "C3, Direct window (round 6 $L)" --config c3 --only-headline > /dev/null 2>&1
rm -rf gpurun_out/prof_round6_${L}_*/*/ 2>/dev/null
ls gpurun_out/ | grep round6_${L}
python3 - <<P
import json
d = json.loads(open('gpurun_out/bench_round6_${L}_bench_c2_direct.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, {k: d['config'][k] for k in list(d['config'])[:20]})
print(d['roofline']['frac'], d['roofline']['kernel_ms'], d['type1']['stages_ms'], d['type2']['stages_ms'])
P
