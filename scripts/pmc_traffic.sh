#!/bin/bash
# HBM-side traffic per launch of the dominant kernels at config 2, as MI355X_MICROARCH.md's HBM section
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (kernel trace only), FETCH_SIZE doubled
# (gfx950 tallies 128-byte requests at 64 B), both in KiB.  Writes profiles/pmc_traffic.json keyed to the
# build stamp of libmgs.so; bench.py prints `roofline.traffic` only when the stamp matches.
#   gpurun -- bash scripts/pmc_traffic.sh        (results also under gpurun_out/pmc_r6/)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_r6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for stage in raster_inf raster_inf_q raster_bwd_split project binning; do
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
    tag=$(echo $ctr | cut -d' ' -f1)
    rm -rf $OUT/${stage}_$tag
    # (the backward is measured on the scene in the caller's order, as bench.py's fwd_bwd leg runs it)
    MORTON=$([ $stage = raster_bwd_split ] && echo 0 || echo 1) SEG=256 timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/${stage}_$tag -o pmc -- python $REPO/scripts/run_stage.py $stage 3 > /dev/null 2>&1
  done
done
python - <<PY
import csv, collections, json, os, sys, glob
sys.path.insert(0, "$REPO")
from robosimgs_amd.csrc import build as hip_build
out = "$OUT"
def agg(d):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out, d, "**", "pmc_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return a
rec = {"stamp": hip_build.current_stamp(), "workload": "configs[1]: 1M Gaussians, SH 3, 1920x1080, tight lists, 4 channels (RGB+ED)", "kernels": {}, "raw": {}}
pick = {"raster_inf": ("raster_fwd", "raster_fwd_kernel<4, false, false>"), "raster_inf_q": ("raster_fwd_q", "raster_fwd_q_kernel<4, false, false>"), "raster_bwd_split": ("raster_bwd", "raster_bwd_kernel"), "project": ("project", "project_color_fwd_kernel")}
for stage in ("raster_inf", "raster_inf_q", "raster_bwd_split", "project", "binning"):
    F, Wr, S = agg(stage + "_FETCH_SIZE"), agg(stage + "_WRITE_SIZE"), agg(stage + "_SQ_INSTS_VALU")
    for k in set(F) | set(Wr) | set(S):
        if "mgs" not in k: continue
        m = lambda a, c: (sum(a[k][c][1:]) / max(1, len(a[k][c][1:]))) if k in a and c in a[k] and len(a[k][c]) > 1 else (a[k][c][0] if k in a and c in a[k] else None)
        row = {"FETCH_SIZE_KiB": m(F, "FETCH_SIZE"), "WRITE_SIZE_KiB": m(Wr, "WRITE_SIZE")}
        for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "GRBM_GUI_ACTIVE"):
            row[c] = m(S, c)
        rec["raw"][stage + ":" + k.replace("void ", "").replace("mgs::(anonymous namespace)::", "")[:70]] = row
        if stage in pick and pick[stage][1] in k and row["FETCH_SIZE_KiB"] is not None and row["WRITE_SIZE_KiB"] is not None:
            rec["kernels"][pick[stage][0]] = {
                "kernel": k[:90], "traffic_bytes": int(2 * row["FETCH_SIZE_KiB"] * 1024 + row["WRITE_SIZE_KiB"] * 1024),
                "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes: 2 x %.1f MB (gfx950 counts 128-byte requests at 64 B) + %.1f MB per launch (profiles/pmc_traffic.json, scripts/pmc_traffic.sh)" % (row["FETCH_SIZE_KiB"] * 1024 / 1e6, row["WRITE_SIZE_KiB"] * 1024 / 1e6)}
json.dump(rec, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(rec["kernels"], indent=1))
for k, v in rec["raw"].items():
    if k.startswith("raster_inf") and "kernel<4, false" in k or k.startswith("raster_bwd_split:") and "raster_bwd" in k:
        print(k[:60], {a: (round(b) if b else b) for a, b in v.items()})
PY
