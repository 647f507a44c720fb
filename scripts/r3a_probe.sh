#!/bin/bash
# round 3, GPU call a: f32 MFMA microbenchmark + phase profile of the patch kernel at C3 / C2 (NUFFT_PATCH_PROFILE build)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3a; mkdir -p $O
hipcc -O3 --offload-arch=gfx950 scripts/microbench8.hip -o /tmp/mb8 && /tmp/mb8 > $O/mb8.log 2>&1
P=$PWD/nonuniformffts.jl_amd/libnufft_prof.so
NUFFT_LIB_PATH=$P timeout 600 python scripts/perf_probe.py --n 512 --np 1e8 --m 8 --z c64 --mode poly --reps 2 > $O/c3_prof.log 2>&1
NUFFT_LIB_PATH=$P NUFFT_SPREAD_METHOD=2 timeout 300 python scripts/perf_probe.py --mode poly --reps 3 > $O/c2_patch_prof.log 2>&1
NUFFT_LIB_PATH=$P NUFFT_SPREAD_METHOD=2 timeout 300 python scripts/perf_probe.py --mode poly --reps 3 --z c128 > $O/c128m4_patch_prof.log 2>&1
timeout 600 python scripts/perf_probe.py --n 512 --np 1e8 --m 8 --z c64 --mode poly --reps 2 > $O/c3.log 2>&1
NUFFT_SPREAD_METHOD=2 timeout 300 python scripts/perf_probe.py --mode poly --reps 3 > $O/c2_patch.log 2>&1
timeout 300 python scripts/perf_probe.py --mode poly --reps 3 > $O/c2_tiles.log 2>&1
tail -n 30 $O/*.log
