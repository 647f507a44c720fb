#!/bin/bash
# FETCH_SIZE + time of the interpolation kernels at C2: 16 waves x 128 registers vs 8 waves x 256 registers (development probe)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r3fetch; mkdir -p $O
for lib in libnufft_mi355x.so libnufft_w2.so; do
  export NUFFT_LIB_PATH=$R/nonuniformffts.jl_amd/$lib
  for z in f64 f32 c64 c128; do
    python3 scripts/perf_probe.py --mode poly --reps 3 --z $z 2>&1 | grep -E "t2_interp" | sed "s/^/$lib $z /"
  done
  python3 scripts/perf_probe.py --mode poly --reps 3 --c 3 2>&1 | grep -E "t2_interp" | sed "s/^/$lib c=3 /"
done
cd /tmp && export TMPDIR=/tmp
export NUFFT_LIB_PATH=$R/nonuniformffts.jl_amd/libnufft_w2.so
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/w2 -- python3 $R/scripts/perf_probe.py --mode poly --reps 1 > $O/w2.log 2>&1
cd $O
python3 - <<'PY'
import csv, glob, collections
for d in ['w2']:
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'interp' in r['Kernel_Name']: agg[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(d, k, 'FETCH raw %.3f GB (x2 = %.3f)' % (sum(v)/len(v)*1024/1e9, 2*sum(v)/len(v)*1024/1e9))
PY
find $O -name "*.db" -delete
