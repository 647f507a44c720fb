cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o p -- python $GRAFT_REPO_ROOT/scripts/perf_probe.py --mode poly --reps 5 $@ > /tmp/log.txt 2>&1
grep -E "set_points  " /tmp/log.txt
python3 - <<PY
import csv,glob
for f in glob.glob("/tmp/prof_k/**/*kernel_stats.csv", recursive=True):
    rows=[r for r in csv.DictReader(open(f))]
    for r in rows:
        n=r["Name"]
        if any(k in n for k in ("bin_","patch_column","patch_task","patch_split","tile_work","tile_slices","fill_desc","scan","lookback","zero")): print("   %-60s calls %4s  %8.1f us"%(n[:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
