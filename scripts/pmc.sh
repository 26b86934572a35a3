#!/bin/bash
# usage: pmc.sh <stage> <outdir-name> "<COUNTERS...>"   (one --pmc pass; kernel-trace only)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $3 --output-format csv -d /root/repo/gpurun_out/$2 -o pmc -- python /root/repo/scripts/run_stage.py $1 3 > /dev/null 2>&1
python - <<PY
import csv, collections
rows = list(csv.DictReader(open('/root/repo/gpurun_out/$2/pmc_counter_collection.csv')))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in agg.items():
    if 'mgs' not in k: continue
    print(k, {n: round(sum(v)/len(v), 1) for n, v in c.items()}, 'launches', len(next(iter(c.values()))))
PY
