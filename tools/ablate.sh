#!/bin/bash
# Time bench.py under each experiment variant of the library (profiling aid; variants compute WRONG gradients).
cd "$(dirname "$0")/.."
for tag in "" "$@"; do
  if [ -z "$tag" ]; then lib=scgaussian_amd/libscg_raster.so; else lib=scgaussian_amd/libscg_raster_$tag.so; fi
  echo "== ${tag:-baseline}"
  SCG_LIB_PATH=$PWD/$lib python bench.py --steps 20 --warmup 3 --sustained-steps 0 --no-cpu-baseline --no-full-iteration ${ABLATE_ARGS:---no-s3} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  iters/s', d['value'], 'ms/step', d['ms_per_step'], 'render_ms', d['render_ms'])
print('  stage_ms', d['stage_ms'])
if 's3_forward' in d: print('  s3', d['s3_forward']['render_ms'], d['s3_forward']['stage_ms'])
"
done
