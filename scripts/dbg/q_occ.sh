for v in "" '-DMGS_RASTER_Q_VGPR_CLOBBER="v71"' '-DMGS_RASTER_Q_VGPR_CLOBBER="v79"' '-DMGS_RASTER_Q_VGPR_CLOBBER="v95"' '-DMGS_RASTER_Q_VGPR_CLOBBER="v127"'; do
  MGS_RASTER_FWD_FLAGS="$v" python -c "from robosimgs_amd.csrc import build; build.build(force=True)" > /dev/null 2>&1
  MGS_RASTER_FWD_FLAGS="$v" MGS_USE_DEBUG_LIB=1 MGS_RASTER_OPTS=5 python bench.py --no-cpu-baseline --bwd-steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v] q kernel forced, 3 in flight:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_by_schedule'])"
done
