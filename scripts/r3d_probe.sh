#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "patch_engine_every_instantiation or both_spreading_engines or engine_choice or callbacks_match" 2>&1 | tail -15 > $O/tests.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "dense_point_sets" 2>&1 | tail -15 >> $O/tests.log
timeout 600 python scripts/perf_probe.py --n 512 --np 1e8 --m 8 --z c64 --mode poly --reps 2 > $O/c3.log 2>&1
NUFFT_PATCH_F32ACC=0 timeout 600 python scripts/perf_probe.py --n 512 --np 1e8 --m 8 --z c64 --mode poly --reps 2 > $O/c3_f64acc.log 2>&1
for m in 4 6; do
timeout 300 python scripts/perf_probe.py --n 256 --np 1e7 --m $m --z c64 --mode poly --reps 3 > $O/c64_m$m.log 2>&1
NUFFT_PATCH_F32ACC=0 timeout 300 python scripts/perf_probe.py --n 256 --np 1e7 --m $m --z c64 --mode poly --reps 3 > $O/c64_m${m}_f64acc.log 2>&1
done
cat $O/tests.log; grep -h "t1_spread\|type-1 exec" $O/c*.log
