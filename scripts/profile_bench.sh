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
# SQ counters of the same command (what binds the kernels: bench.py derives roofline.binding_resource from these)
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq_a -- python3 $R/bench.py --steps 3 --warmup 1 --only-headline "$@" > $OUT/sq_a_line.json 2> $OUT/sq_a_stderr.txt
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $OUT/sq_b -- python3 $R/bench.py --steps 3 --warmup 1 --only-headline "$@" > $OUT/sq_b_line.json 2> $OUT/sq_b_stderr.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq_c -- python3 $R/bench.py --steps 3 --warmup 1 --only-headline "$@" > $OUT/sq_c_line.json 2> $OUT/sq_c_stderr.txt
cd $R
# keep only the small csv files (the merged-back directory is capped at 64 MiB)
find $OUT -name "*.db" -delete 2>/dev/null
find $OUT -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
python3 scripts/summarize_profile.py $OUT $R/gpurun_out/summary_$TAG "$NOTE" > /dev/null 2>&1
cat $R/gpurun_out/bench_$TAG.json
