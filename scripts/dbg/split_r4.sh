#!/bin/bash
cd /root/repo
O=gpurun_out/r4e; mkdir -p $O
python -m pytest tests/test_gpu_backward.py -x -q 2>&1 | tail -5 > $O/bwd_tests.txt
SEGS=64,128,256 python scripts/raster_bwd_split_ab.py 2>&1 | grep segment > $O/ab.txt
MGS_RASTER_BWD_FLAGS="-DMGS_RASTER_BWD_TIMING" python robosimgs_amd/csrc/build.py > /dev/null 2>&1
for s in 128 256; do MGS_RASTER_BWD_FLAGS="-DMGS_RASTER_BWD_TIMING" SEG=$s python scripts/dbg/bwd_timeline.py 2>&1 | grep -v amdgpu.ids > $O/timeline_$s.txt; done
python robosimgs_amd/csrc/build.py > /dev/null 2>&1
cat $O/bwd_tests.txt $O/ab.txt $O/timeline_*.txt
