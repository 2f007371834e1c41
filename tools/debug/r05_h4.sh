#!/bin/bash
R=$GRAFT_REPO_ROOT
for w in S2 S3 S4; do
echo "== $w"; bash $R/tools/kstats.sh h4_$w python $R/tools/debug/r05_precomp.py $w | grep "geometry_hist\|scatter\|colscan\|tile_blend"
tail -3 $R/gpurun_out/h4_$w/cmd.log | cut -c1-300
rm -f $R/gpurun_out/h4_$w/*trace*.csv
done
