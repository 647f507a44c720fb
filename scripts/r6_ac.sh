#!/bin/bash
# the N > 1 launch the driver uses, two ranks sharing the one GPU of this box (NUFFT_BENCH_SHARE_GPU=1): does the line still come out, with the lead keys
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6ac}
export NUFFT_BENCH_SHARE_GPU=1
timeout 1500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > $O/${TAG}_line.json 2> $O/${TAG}_stderr.txt; echo "rc=$?"
tail -3 $O/${TAG}_stderr.txt
python3 - <<PY
import json
l = [x for x in open('gpurun_out/${TAG}_line.json').read().strip().splitlines() if x.startswith('{')]
print(len(l), 'json lines')
d = json.loads(l[-1])
print({k: d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','scaling','dtype','vs_baseline')})
print(list(d['config'])[:8])
PY
