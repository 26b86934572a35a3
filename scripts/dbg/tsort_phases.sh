#!/bin/bash
# instructions of the per-tile sort by phase: the kernel cut short after phase k (wrong lists: measurement only)
for k in ${PHASES:-1 2 3 4 0}; do
  MGS_TILE_SORT_FLAGS="-DMGS_TSORT_STOP=$k $EXTRA" python robosimgs_amd/csrc/build.py --force > /dev/null 2>&1 || echo BUILD FAILED
  echo "STOP=$k $EXTRA"
  MGS_TILE_SORT_FLAGS="-DMGS_TSORT_STOP=$k $EXTRA" timeout 120 bash scripts/pmc.sh binning r3/pmc_ts_$k "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" 2>&1 | grep "depth_sort" | sed 's/.*{/{/'
done
