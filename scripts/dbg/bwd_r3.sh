#!/bin/bash
# round 3: backward raster A/B on the GPU box.  usage: bwd_r3.sh "<flags A>" "<flags B>" ...
for flags in "$@"; do
  MGS_RASTER_BWD_FLAGS="$flags" python robosimgs_amd/csrc/build.py --force > /dev/null 2>&1 || echo BUILD FAILED
  MGS_RASTER_BWD_FLAGS="$flags" TAG="[$flags]" python scripts/raster_bwd_ab.py 2>&1 | tail -1
done
