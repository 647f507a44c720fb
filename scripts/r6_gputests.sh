#!/bin/bash
# GPU parity suite + smoke on the tree as it stands (the driver's round-end check, run early)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
TAG=${1:-r6}
python3 -m pytest tests -m gpu -q -p no:cacheprovider > $O/${TAG}_gputests.txt 2>&1
echo "pytest rc=$?" >> $O/${TAG}_gputests.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.txt 2>&1
echo "smoke rc=$?" >> $O/${TAG}_smoke.txt
./tests/c_abi_smoke > $O/${TAG}_cabi.txt 2>&1; echo "c_abi rc=$?" >> $O/${TAG}_cabi.txt
tail -5 $O/${TAG}_gputests.txt; tail -3 $O/${TAG}_smoke.txt; tail -2 $O/${TAG}_cabi.txt
