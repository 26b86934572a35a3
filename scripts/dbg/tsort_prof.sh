cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ts_prof -o s -- python $GRAFT_REPO_ROOT/scripts/binning_ab.py 2>&1 | tail -1
python - <<PY
import csv, glob, re
for f in glob.glob("/tmp/ts_prof/**/s_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "mgs" in n:
            n = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0]
            print("%-60s calls %5s avg %7.1f us" % (n[-58:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf /tmp/ts_prof
