#!/bin/bash
# Where the backward raster's HBM-side traffic comes from: FETCH_SIZE / WRITE_SIZE (separate --pmc passes, kernel trace only)
# of raster_bwd_kernel for the scene in the caller's order and in Morton order, whole-list walk and 256-entry segments.
#   gpurun -- bash scripts/pmc_bwd_matrix.sh        (results under gpurun_out/pmc_bwd/)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_bwd
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for morton in 0 1; do
  for stage in raster_bwd_det raster_bwd_split; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
      d=$OUT/${stage}_m${morton}_$ctr
      rm -rf $d
      MORTON=$morton SEG=256 timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $d -o pmc -- python $REPO/scripts/run_stage.py $stage 3 > /dev/null 2>&1
    done
  done
done
python - <<PY
import csv, collections, glob, os
out = "$OUT"
print("| scene order | walk | kernel | FETCH_SIZE MB (raw) | x 2 | WRITE_SIZE MB | 2 x FETCH + WRITE |")
print("|---|---|---|---:|---:|---:|---:|")
for morton in (0, 1):
    for stage in ("raster_bwd_det", "raster_bwd_split"):
        vals = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            a = collections.defaultdict(list)
            for f in glob.glob(os.path.join(out, f"{stage}_m{morton}_{ctr}", "**", "pmc_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == ctr:
                        a[r["Kernel_Name"]].append(float(r["Counter_Value"]))
            for k, v in a.items():
                if "raster_bwd_kernel" in k or "reduce_records" in k:
                    name = "raster_bwd_kernel" if "raster_bwd_kernel" in k else "reduce_records_rows_kernel"
                    vals.setdefault(name, {})[ctr] = sum(v[1:]) / max(1, len(v[1:])) * 1024 / 1e6
        for name, d in vals.items():
            f_, w_ = d.get("FETCH_SIZE", 0.0), d.get("WRITE_SIZE", 0.0)
            print(f"| {'Morton' if morton else 'as given (random)'} | {'whole list' if stage == 'raster_bwd_det' else 'segments of 256'} | {name} | {f_:.1f} | {2 * f_:.1f} | {w_:.1f} | {2 * f_ + w_:.1f} |")
PY
