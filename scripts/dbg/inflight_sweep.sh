for f in 2 3 4 5 6; do
  python bench.py --no-cpu-baseline --inflight $f --bwd-steps 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('in flight $f: frames/s', d['value'])"
done
MGS_USE_DEBUG_LIB=1 MGS_RASTER_OPTS=7 python bench.py --no-cpu-baseline --bwd-steps 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('per-block kernel, 3 in flight: frames/s', d['value'])"
