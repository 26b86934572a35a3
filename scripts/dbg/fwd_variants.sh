# rebuild raster_fwd.hip with different workgroup shapes / register budgets of the per-tile kernel; 3-in-flight bench
for v in "" "-DMGS_RASTER_WG_WAVES=2" "-DMGS_RASTER_WG_WAVES=8" "-DMGS_RASTER_WG_WAVES=1" "-DMGS_RASTER_WAVES=5" "-DMGS_RASTER_WAVES=3"; do
  MGS_RASTER_FWD_FLAGS="$v" python -c "from robosimgs_amd.csrc import build; build.build(force=True)" > /dev/null 2>&1
  MGS_RASTER_FWD_FLAGS="$v" python bench.py --no-cpu-baseline --bwd-steps 2 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$v] frames/s', d['value'], d['roofline']['kernel_ms_by_schedule'])"
done
