O=gpurun_out/r05k; mkdir -p $O
tools/pmc.sh r05k/pmc > $O/pmc_counters.txt 2>&1; tail -14 $O/pmc_counters.txt
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
