#!/bin/bash
# forward raster variants: kernel alone (raster_ab.py) and frames/s with three in flight (bench.py).  usage: fwd_fps.sh "<flags A>" ...
for flags in "$@"; do
  echo "=== MGS_RASTER_FWD_FLAGS='$flags'"
  MGS_RASTER_FWD_FLAGS="$flags" python robosimgs_amd/csrc/build.py --force > /dev/null 2>&1 || echo BUILD FAILED
  MGS_RASTER_FWD_FLAGS="$flags" python scripts/raster_ab.py 1 5 2>&1 | grep "opts="
  MGS_RASTER_FWD_FLAGS="$flags" python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('frames/s', d['value'], 'given order', c.get('frames_per_s_scene_in_given_order'), 'latency ms', c.get('single_frame_latency_ms'), 'train ms', [v for k, v in c.items() if 'train' in k])"
done
