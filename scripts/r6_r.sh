#!/bin/bash
R=$(pwd); O=$R/gpurun_out; TAG=${1:-r6r}
timeout 600 python3 -m pytest tests -m gpu -q -x -p no:cacheprovider -k "adaptive_sort" > $O/${TAG}_tests.txt 2>&1; tail -30 $O/${TAG}_tests.txt | cut -c1-220
