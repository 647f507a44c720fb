#!/bin/bash
# usage: scripts/prof_probe.sh <perf_probe args...>  -> per-kernel stats of one perf_probe run
R=$(pwd); OUT=$R/gpurun_out/prof_probe; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/scripts/perf_probe.py --reps 3 "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.05:
        print("%-90s calls %4s avg %9.1f us  %5.2f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
