# launch order of the raster kernels: everywhere (default) / not in the per-tile inference kernel (16) / nowhere (8)
for o in 3 19 11; do
  echo "== MGS_USE_DEBUG_LIB=1 MGS_RASTER_OPTS=$o"
  MGS_USE_DEBUG_LIB=1 MGS_RASTER_OPTS=$o python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   frames/s', d['value'], 'single', d['config']['single_frame_latency_ms'], 'fwd_bwd', d['fwd_bwd']['ms_per_step'], d['roofline']['kernel_ms_by_schedule'])"
  MGS_USE_DEBUG_LIB=1 MGS_RASTER_OPTS=$o python bench.py --no-cpu-baseline --inflight 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   inflight1 frames/s', d['value'])"
done
python scripts/raster_bwd_ab.py | tail -1
