# rebuild libmgs.so on the GPU box with different raster_bwd flags and time each
for v in "" "-DMGS_RASTER_BWD_ORDER=0" "-DMGS_RASTER_BWD_PIPE=0" "-DMGS_RASTER_BWD_ORDER=0 -DMGS_RASTER_BWD_PIPE=0" "-DMGS_RASTER_BWD_MIN_WAVES=5"; do
  MGS_RASTER_BWD_FLAGS="$v" python -c "from robosimgs_amd.csrc import build; build.build(force=True)" > /dev/null 2>&1
  TAG="[$v]" MGS_RASTER_BWD_FLAGS="$v" python scripts/raster_bwd_ab.py 2>&1 | tail -1
done
