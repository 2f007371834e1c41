#!/bin/bash
# A/B on the same box: bench with and without an environment switch.  Usage: tools/_ab.sh VAR [bench args]
var=$1; shift
for rep in 1 2; do
for v in "" 1; do
  if [ -z "$v" ]; then unset $var; else export $var=1; fi
  python bench.py --steps 40 --warmup 5 --sustained-steps 0 --no-cpu-baseline --no-full-iteration "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$var=${v:-0}', d['value'], d['ms_per_step'], d['stage_ms'], d.get('s3_forward',{}).get('stage_ms'))"
done; done
