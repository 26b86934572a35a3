#!/bin/bash
# where does reduce_records_kernel's time go: variants without record loads / flag loads / stores, kernel time from a trace
for e in 0 1 2 3; do
  MGS_RASTER_BWD_FLAGS="-DMGS_REDUCE_EXP=$e" python robosimgs_amd/csrc/build.py --force > /dev/null 2>&1 || echo BUILD FAILED
  echo "== MGS_REDUCE_EXP=$e"
  MGS_RASTER_BWD_FLAGS="-DMGS_REDUCE_EXP=$e" bash scripts/dbg/train_trace.sh 2>&1 | grep "reduce_records\|raster_bwd_kernel"
done
