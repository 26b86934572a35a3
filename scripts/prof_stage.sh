#!/bin/bash
# usage: prof_stage.sh <stage> <reps>  -> per-kernel stats table for one stage
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/ps_$1
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/ps_$1 -o s -- python /root/repo/scripts/run_stage.py $1 ${2:-10} > /dev/null 2>&1
python /root/repo/scripts/summarize_rocprof.py /root/repo/gpurun_out/ps_$1/s_kernel_stats.csv 14 | grep -v "at::native"
