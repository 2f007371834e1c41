for v in 1 0; do
  echo "== SCG_FUSED_SORT=$v"
  SCG_FUSED_SORT=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-full-iteration --no-small 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['render_ms']); print(d['stage_ms']); print(d['stage_ms_forward_only']); print(d['s3_forward']['render_ms'], d['s3_forward']['stage_ms'])
"
done
