#!/bin/bash
cd /root/repo
for v in "8" "4"; do
  F="-DMGS_DIRECT_PER_THREAD=$v"
  MGS_BINNING_FLAGS="$F" python robosimgs_amd/csrc/build.py > /dev/null 2>&1
  for m in 1 0; do
    cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/dsp
    MGS_BINNING_FLAGS="$F" MORTON=$m timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dsp -o s -- python /root/repo/scripts/dbg/tsort_time.py > /dev/null 2>&1
    echo "== per thread $v, morton=$m"
    python /root/repo/scripts/summarize_rocprof.py $(find /tmp/dsp -name "s_kernel_stats.csv" | head -1) 12 | grep -E "direct_|tile_depth" | cut -c1-110
    cd /root/repo
  done
done
python robosimgs_amd/csrc/build.py > /dev/null 2>&1
