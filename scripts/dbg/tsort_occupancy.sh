#!/bin/bash
# Waves of the per-tile sort's main kernel resident on average, from SQ_WAVE_CYCLES (quad-cycles) and SQ_BUSY_CYCLES (summed over 32
# shader engines) of one rocprofv3 --pmc pass, at configs[4] and configs[1]:   gpurun -- bash scripts/dbg/tsort_occupancy.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/tsort_occ
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
N=5000000 MU=0.008 W=3840 H=2160 CAP=30100000 timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/4k -o pmc -- python $REPO/scripts/run_stage.py binning 5 > $OUT/4k.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/1080 -o pmc -- python $REPO/scripts/run_stage.py binning 5 > $OUT/1080.log 2>&1
python - <<PY
import csv, collections, glob, os
for which in ("4k", "1080"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join("$OUT", which, "**", "pmc_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "mgs" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        a = {n: sum(v) / len(v) for n, v in c.items()}
        if a.get("SQ_BUSY_CYCLES"):
            print(which, k, "waves %d, resident on average %.0f (%.1f per CU)" % (a["SQ_WAVES"], 4 * a["SQ_WAVE_CYCLES"] / (a["SQ_BUSY_CYCLES"] / 32), 4 * a["SQ_WAVE_CYCLES"] / (a["SQ_BUSY_CYCLES"] / 32) / 256))
PY
