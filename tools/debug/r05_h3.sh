#!/bin/bash
# forward blend: fewer workgroups per compute unit (LDS pad) so that S2's 3 024 tiles make whole rounds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/h3
timeout 900 python tools/ab_inproc.py --workload S2 --mode render --libs ,p7,p6,p5 --reps 4 --steps 150 --warm 400 > gpurun_out/h3/ab_S2r.txt 2>&1
tail -10 gpurun_out/h3/ab_S2r.txt
timeout 900 python tools/ab_inproc.py --workload S2 --libs ,p7,p6 --reps 4 --steps 150 --warm 400 > gpurun_out/h3/ab_S2.txt 2>&1
tail -8 gpurun_out/h3/ab_S2.txt
timeout 900 python tools/ab_inproc.py --workload S3 --mode render --libs ,p7,p6 --reps 3 --steps 100 --warm 300 > gpurun_out/h3/ab_S3r.txt 2>&1
tail -8 gpurun_out/h3/ab_S3r.txt
