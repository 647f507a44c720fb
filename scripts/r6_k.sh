#!/bin/bash
# dense engine after: unconditional pipeline loads, no LDS queue drains between strip writes and reads, flush table reads hoisted
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6k}; P=$O/${TAG}_probes.txt; : > $P
timeout 900 python3 -m pytest tests -m gpu -q -x -p no:cacheprovider -k "dense_window" > $O/${TAG}_tests.txt 2>&1; tail -3 $O/${TAG}_tests.txt
run() { echo "=== $*" >> $P; env "${ENVV[@]}" python3 scripts/perf_probe.py --z f64 --reps 4 "$@" 2>&1 | grep -E "t1_spread" >> $P; }
for mode in direct poly; do
  for cfg in "--m 4 --sigma 1.5 --np 16777216" "--m 4 --sigma 1.5 --np 5305433" "--m 4 --sigma 1.5 --np 53054326" "--m 4 --sigma 1.5 --np 16777216 --dist randn" "--m 4 --sigma 1.5 --np 53054326 --dist randn" "--m 4" "--m 5" "--m 6" "--m 3 --np 4e7" "--m 2 --np 4e7" "--m 5 --np 4e7" "--m 6 --np 4e7"; do
    ENVV=(NUFFT_DENSE_MIN=0); run --mode $mode $cfg
    ENVV=(NUFFT_DENSE=0); run --mode $mode $cfg
  done
done
python3 - <<'PY'
import re
txt=open('gpurun_out/r6k_probes.txt').read()
rows=re.split(r'^=== ',txt,flags=re.M)[1:]
out=[]
for r in rows:
    m=re.search(r't1_spread\s+([\d.]+)',r); out.append((r.split('\n')[0], float(m.group(1)) if m else None))
for k in range(0,len(out)-1,2):
    a,b=out[k],out[k+1]
    print(f"{a[0]:72s} dense {a[1]:8.3f} atomic {b[1]:8.3f} ratio {a[1]/b[1]:.2f}")
PY
