#!/bin/bash
# per-kernel rocprofv3 times of the binning kernels for column-scan shapes (bins per workgroup), one frame at a time
cd /root/repo
for b in 16 8 4 32; do
  MGS_BINNING_FLAGS="-DMGS_COLSCAN_BINS=$b" python robosimgs_amd/csrc/build.py > /dev/null 2>&1
  MGS_BINNING_FLAGS="-DMGS_COLSCAN_BINS=$b" bash scripts/profile_bench.sh colscan_$b --inflight 1 --no-cpu-baseline --no-stress > /dev/null 2>&1
  echo "== bins per workgroup $b"; grep -E "direct_|tile_depth_sort_kernel<true, 1024>" gpurun_out/colscan_${b}_stats.md | cut -c1-110
done
python robosimgs_amd/csrc/build.py > /dev/null 2>&1
