# bench.py processes alternating between the product library and libscg_raster_prev.so (200 steps each, three rounds):
# ms per step WITH the two stage events around blend_backward | sustained | blend_backward event time | render.  Run on the GPU box.
for i in 1 2 3; do
for lib in "" prev; do
  if [ -n "$lib" ]; then export SCG_LIB_PATH=$PWD/scgaussian_amd/libscg_raster_$lib.so; else unset SCG_LIB_PATH; fi
  python bench.py --steps 200 --warmup 50 --no-s3 --no-small --no-clustered --no-cpu-baseline --no-full-iteration --no-rccl-floor 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('lib=${lib:-new}', d['ms_per_step'], d['sustained']['ms_per_step'], d['roofline']['mean_ms'], d.get('render_ms'))"
done; done
