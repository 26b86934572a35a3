#!/bin/bash
# per-tile sort: workgroup size x buckets (lists must stay bit-identical: the forward list tests run on each build)
cd /root/repo
for v in "256 1024" "128 1024" "128 512" "256 512" "512 1024"; do
  set -- $v
  F="-DMGS_TSORT_THREADS=$1 -DMGS_TSORT_BUCKETS=$2"
  MGS_TILE_SORT_FLAGS="$F" python robosimgs_amd/csrc/build.py > /dev/null 2>&1 || echo BUILD FAILED
  ok=$(MGS_TILE_SORT_FLAGS="$F" timeout 300 python -m pytest tests/test_gpu_forward.py -x -q -k "isect or depth_order or partition or binning" 2>&1 | tail -1)
  MGS_TILE_SORT_FLAGS="$F" TAG="threads $1 buckets $2 [$ok]" timeout 120 python scripts/dbg/tsort_time.py 2>&1 | grep binning
done
python robosimgs_amd/csrc/build.py > /dev/null 2>&1
