#!/bin/bash
# frames in flight x GPU_MAX_HW_QUEUES (HIP maps streams onto that many hardware queues; default 4)
cd /root/repo
for q in 4 8; do for f in 2 3 4 5 6; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --no-stress --bwd-steps 2 --inflight $f 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('hw queues $q, in flight $f:', r['value'], 'given order', r['config']['frames_per_s_scene_in_given_order'], 'latency', r['config']['single_frame_latency_ms'])"
done; done
