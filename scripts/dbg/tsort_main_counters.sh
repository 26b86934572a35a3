#!/bin/bash
# Where the per-tile sort's main kernel spends configs[4]'s extra ~10 us against the round-5 library: the same 4K binning stage
# under rocprofv3 --pmc (kernel trace only, one pass per counter group) with each library in turn.
#   gpurun -- bash scripts/dbg/tsort_main_counters.sh <round-5 libmgs.so>      (results under gpurun_out/tsort_ctr/)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/tsort_ctr
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BASE=$(realpath $REPO/$1)
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_WAVES SQ_ACTIVE_INST_ANY"; do
  i=$((i + 1))
  for which in cur base; do
    if [ $which = base ]; then export VARIANT_LIB=$BASE; else unset VARIANT_LIB; fi
    N=5000000 MU=0.008 W=3840 H=2160 CAP=30100000 timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/${which}_$i -o pmc -- python $REPO/scripts/run_stage.py binning 5 > $OUT/${which}_$i.log 2>&1
  done
done
python - <<PY
import csv, collections, glob, os
out = "$OUT"
for which in ("base", "cur"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out, which + "_*", "**", "pmc_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "tile_depth_sort" in r["Kernel_Name"] or "unit_" in r["Kernel_Name"] or "sort_units" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        print(which, k, {n: round(sum(v) / len(v)) for n, v in sorted(c.items())})
PY
