#!/bin/bash
# same box: stage times with an environment switch off / on.  usage: ab_env.sh VAR [value_off value_on]
var=$1; off=${2:-0}; on=${3:-1}
for rep in 1 2; do
for v in $off $on; do
  export $var=$v
  python bench.py --steps 40 --warmup 10 --sustained-steps 0 --no-cpu-baseline --no-full-iteration --no-small 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$v', d['ms_per_step'], d['stage_ms'], 'S3', d['s3_forward']['render_ms'], d['s3_forward']['stage_ms'])"
done; done
