#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$PWD/gpurun_out/r3calib; mkdir -p $O
hipcc -O3 --offload-arch=gfx950 scripts/fetch_calibration.hip -o /tmp/fetch_calibration || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- /tmp/fetch_calibration > $O/stdout.txt 2> $O/stderr.txt
cd $O
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('fetch/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name']].append(float(r['Counter_Value']))
with open('fetch_calibration.txt', 'w') as out:
    out.write(open('stdout.txt').read())
    for k, v in agg.items():
        out.write('%-70s launches %d  FETCH_SIZE avg %.4f GB (KiB x 1024)\n' % (k[:70], len(v), sum(v) / len(v) * 1024 / 1e9))
print(open('fetch_calibration.txt').read())
PY
find $O -name "*.db" -delete
