#!/bin/bash
# the round's last pass: the collection on the final kernels, then the quality proxy, the scenario sweep and the soak
cd $GRAFT_REPO_ROOT
FUZZ_N=300 FUZZ_F=500 bash tools/collect_round.sh r05z
O=gpurun_out/r05z
timeout 1500 python tools/quality_proxy.py --iters 500 --json $O/quality_proxy_504x378_500its.json > $O/quality_proxy.log 2>&1
timeout 900 python tools/scenario_sweep.py > $O/scenario_sweep.txt 2>&1
timeout 900 python tools/soak.py 20000 > $O/soak_20000.txt 2>&1
tail -2 $O/quality_proxy.log; tail -2 $O/soak_20000.txt
