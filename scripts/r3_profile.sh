#!/bin/bash
# round 3 profiles: the committed bench line (C2 + compact C3 / C4 records), then rocprofv3 kernel stats and PMC traffic
# of C2 (polynomial and Direct windows separately), C3 and C4.  usage (through gpurun): bash scripts/r3_profile.sh <letter>
L=${1:-a}
cd "$GRAFT_REPO_ROOT" || exit 1
bash scripts/profile_bench.sh round3_${L}_bench_c2 "C2, FastApproximation window (round 3 $L)" > /dev/null 2>&1
bash scripts/profile_bench.sh round3_${L}_bench_c2_direct "C2, Direct window (the ROC default; round 3 $L)" --evalmode direct --only-headline > /dev/null 2>&1
bash scripts/profile_bench.sh round3_${L}_bench_c3 "C3 (round 3 $L)" --config c3 --only-headline > /dev/null 2>&1
bash scripts/profile_bench.sh round3_${L}_bench_c4 "C4 (round 3 $L)" --config c4 --only-headline > /dev/null 2>&1
ls gpurun_out/ | grep round3_${L}
head -c 3000 gpurun_out/bench_round3_${L}_bench_c2.json
