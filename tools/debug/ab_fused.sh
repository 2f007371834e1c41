#!/bin/bash
# same-process A/B of the forward blend that sorts its own tiles vs sort kernel + blend kernel (module switch of the binding)
cd "$(dirname "$0")/../.."
for w in ${1:-S2 S3 S4}; do python tools/ab_inproc.py --workload $w --mode render --switch FUSED_SORT=True,False --reps 4; done
