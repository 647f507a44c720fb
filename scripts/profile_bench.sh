#!/bin/bash
# usage (on the GPU box, through gpurun): scripts/profile_bench.sh <tag> <note> [bench.py arguments, e.g. --config c3]
# Runs bench.py plainly (the committed bench line), then under rocprofv3 --kernel-trace --stats, then two separate
# PMC passes (FETCH_SIZE, WRITE_SIZE: never combined with trace domains other than --kernel-trace).  Raw output goes
# to gpurun_out/prof_<tag>/; scripts/summarize_profile.py condenses it into gpurun_out/summary_<tag>* (copy those
# into profiles/).
TAG=${1:-r2}; NOTE=${2:-}; shift; shift
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 bench.py "$@" > $R/gpurun_out/bench_$TAG.json 2> $OUT/bench_stderr.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 10 --warmup 2 --only-headline "$@" > $OUT/stats_line.json 2> $OUT/stats_stderr.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --only-headline "$@" > $OUT/fetch_line.json 2> $OUT/fetch_stderr.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 3 --warmup 1 --only-headline "$@" > $OUT/write_line.json 2> $OUT/write_stderr.txt
cd $R
# keep only the small csv files (the merged-back directory is capped at 64 MiB)
find $OUT -name "*.db" -delete 2>/dev/null
find $OUT -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
python3 scripts/summarize_profile.py $OUT $R/gpurun_out/summary_$TAG "$NOTE" > /dev/null 2>&1
cat $R/gpurun_out/bench_$TAG.json
