cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tt -o t -- python $GRAFT_REPO_ROOT/scripts/dbg/train_trace.py > /dev/null 2>&1
python - <<PY
import csv, glob, re
rows = []
for f in glob.glob("/tmp/tt/**/t_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:70]))
for f in glob.glob("/tmp/tt/**/t_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY " + r.get("Direction", "")))
rows.sort()
# last step = everything after the last project_color_fwd launch
idx = max(i for i, r in enumerate(rows) if "project_color_fwd" in r[2])
t0 = rows[idx][0]
prev = t0
for s, e, nme in rows[idx:]:
    print("%8.1f us  +gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, nme))
    prev = e
PY
rm -rf /tmp/tt
