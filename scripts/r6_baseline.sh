#!/bin/bash
# round 6, first GPU call: the driver-style bench line of the tree as committed (are C3 / C4 among the first 20 scalars of config?),
# and the round-5 state of the plans that review item 3 names (set_points / interpolation of Float32 m = 4, Float64 m = 6, ComplexF64 m = 6)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
python3 bench.py > $O/r6_a_bench_line.json 2> $O/r6_a_bench_stderr.txt
python3 - <<'P' > $O/r6_a_first20.txt
import json, sys
sys.path.insert(0, '.')
import importlib.util
spec = importlib.util.spec_from_file_location("b", "bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
d = json.loads(open('gpurun_out/r6_a_bench_line.json').read().strip().splitlines()[-1])
for k in b.first_scalar_keys(d['config'], 20): print(k, d['config'][k])
print('value', d['value'], 'roofline.frac', d['roofline']['frac'], 'interp', {k: d['roofline']['interp'].get(k) for k in ('kernel','kernel_ms','profile_kernel_us','profile_matches_run','traffic_source')})
P
for cfg in "f32 4" "f64 6" "c128 6" "f64 4" "c128 4" "f32 6" "f64 5" "f64 3" "f64 2"; do
  set -- $cfg
  for mode in direct poly; do
    echo "=== z=$1 m=$2 mode=$mode" >> $O/r6_a_probes.txt
    python3 scripts/perf_probe.py --z $1 --m $2 --mode $mode --reps 5 >> $O/r6_a_probes.txt 2>&1
  done
done
cat $O/r6_a_first20.txt
