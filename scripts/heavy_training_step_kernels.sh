#!/bin/bash
# per-kernel rocprofv3 averages of the training step on the heavy-tailed scene, Morton order and as given (profiles/r5/18_*.txt):  gpurun -- bash scripts/heavy_training_step_kernels.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for order in 1 0; do
rm -rf /tmp/prof_t; SCENE=heavy MORTON=$order timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o s -- python $R/scripts/run_stage.py train 6 > /tmp/t.out 2>&1
echo "## heavy-tailed scene, training step kernels, MORTON=$order"; tail -2 /tmp/t.out
python - <<PY
import csv, glob
for f in glob.glob("/tmp/prof_t/**/s_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mgs" in r["Name"]: print("%-75s calls %4s avg %8.1f us" % (r["Name"].replace("void ","").replace("mgs::(anonymous namespace)::","")[:75], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
