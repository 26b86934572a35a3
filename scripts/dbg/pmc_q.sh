cd /tmp && export TMPDIR=/tmp
for o in 1 3; do
rm -rf /root/repo/gpurun_out/pmcq_$o
MGS_USE_DEBUG_LIB=1 MGS_RASTER_OPTS=$o timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d /root/repo/gpurun_out/pmcq_$o -o pmc -- python /root/repo/scripts/run_stage.py raster_inf 3 > /dev/null 2>&1
python - <<PY
import csv, collections, glob
a = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/root/repo/gpurun_out/pmcq_$o/**/pmc_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "raster_fwd" in r["Kernel_Name"] and "false>" in r["Kernel_Name"]:
            a[r["Kernel_Name"][40:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in a.items():
    print("opts=$o", k, {n: round(sum(v[1:]) / max(1, len(v[1:])) / 1e6, 2) for n, v in c.items()})
PY
done
