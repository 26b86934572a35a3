#!/bin/bash
# round 3: forward raster A/B on the GPU box.  usage: fwd_r3.sh "<flags A>" "<flags B>" ...
mkdir -p gpurun_out/r3
for flags in "$@"; do
  echo "=== MGS_RASTER_FWD_FLAGS='$flags'"
  MGS_RASTER_FWD_FLAGS="$flags" python robosimgs_amd/csrc/build.py --force > /dev/null 2>&1 || echo BUILD FAILED
  MGS_RASTER_FWD_FLAGS="$flags" python scripts/raster_ab.py 1 5 2>&1 | grep -v amdgpu.ids
done
