#!/bin/bash
# last check of the tree as committed: smoke, C-ABI driver, the transform tests, one C2 probe
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6an}
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
./tests/c_abi_smoke 2>&1 | tail -1
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "type1_type2 or fftshift or oversampled_grid or halo_variant_every" 2>&1 | tail -2
python3 scripts/perf_probe.py --z f64 --m 4 --np 1e7 --mode direct --reps 8 2>&1 | grep -E "set_points |t1_|t2_|with set_points"
