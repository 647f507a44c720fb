#!/bin/bash
# usage: scripts/pmc_probe.sh <outdir> <perf_probe args...>   (run on the GPU box through gpurun)
OUT=$1; shift
R=$(pwd)
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $R/$OUT/a -- python $R/scripts/perf_probe.py --reps 2 "$@" > $R/$OUT/log_a.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $R/$OUT/b -- python $R/scripts/perf_probe.py --reps 2 "$@" > $R/$OUT/log_b.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM --output-format csv -d $R/$OUT/c -- python $R/scripts/perf_probe.py --reps 2 "$@" > $R/$OUT/log_c.txt 2>&1
python3 - <<PY
import csv, collections, glob
for sub in ("a","b","c"):
    for f in glob.glob("$R/$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in agg.items():
            if 'tile_kernel' in k or 'bin_' in k or 'patch' in k or 'gather' in k or 'march' in k:
                print(k)
                for c,vals in sorted(v.items()):
                    print("   %-24s %.4g"%(c,sum(vals)/len(vals)))
PY
