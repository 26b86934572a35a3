#!/bin/bash
# usage: profile_bench.sh <tag> [bench args...]  -> gpurun_out/<tag>_stats.md + <tag>_line.json
# rocprofv3 --kernel-trace --stats around bench.py (kernel trace only: no PMC in this pass)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_$tag -o s -- python /root/repo/bench.py "$@" > /root/repo/gpurun_out/${tag}_stdout.log 2>&1
grep "^{\"metric\"" /root/repo/gpurun_out/${tag}_stdout.log | tail -1 > /root/repo/gpurun_out/${tag}_line.json
python /root/repo/scripts/summarize_rocprof.py /root/repo/gpurun_out/prof_$tag/s_kernel_stats.csv 34 > /root/repo/gpurun_out/${tag}_stats.md
rm -f /root/repo/gpurun_out/prof_$tag/s_kernel_trace.csv
