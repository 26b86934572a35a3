cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ts_prof -o s -- python $GRAFT_REPO_ROOT/scripts/binning_ab.py 2>&1 | tail -1
python - <<PY
import csv, glob
for f in glob.glob("/tmp/ts_prof/**/s_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "mgs" in n:
            print("%-60s calls %5s avg %7.1f us" % (n.split("(")[0][-58:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf /tmp/ts_prof
