#!/bin/bash
cd /root/repo
for k in 1 2 3 4 0; do
  MGS_TILE_SORT_FLAGS="-DMGS_TSORT_STOP=$k" python robosimgs_amd/csrc/build.py > /dev/null 2>&1 || echo BUILD FAILED
  MGS_TILE_SORT_FLAGS="-DMGS_TSORT_STOP=$k" TAG="sort cut after phase $k (0 = whole)" timeout 120 python scripts/dbg/tsort_time.py 2>&1 | grep binning
done
python robosimgs_amd/csrc/build.py > /dev/null 2>&1
