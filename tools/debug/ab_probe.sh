#!/bin/bash
# same box: stage times for variant libraries (tools/debug: experiments only).  usage: ab_probe.sh tag [tag...]
for rep in 1 2; do
for tag in "" "$@"; do
  if [ -z "$tag" ]; then unset SCG_LIB_PATH; else export SCG_LIB_PATH=$PWD/scgaussian_amd/libscg_raster_$tag.so; fi
  python bench.py --steps 40 --warmup 10 --sustained-steps 0 --no-cpu-baseline --no-full-iteration --no-small 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tag=${tag:-base}', d['ms_per_step'], d['stage_ms'], 'S3', d['s3_forward']['render_ms'], d['s3_forward']['stage_ms'])"
done; done
