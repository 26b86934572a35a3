#!/bin/bash
cd /root/repo; O=gpurun_out/r4m; mkdir -p $O
for r in 1 2 4 8; do
  MGS_RASTER_BWD_FLAGS="-DMGS_RASTER_BWD_XCD_RUN=$r" python robosimgs_amd/csrc/build.py > /dev/null 2>&1
  echo "== XCD_RUN $r" >> $O/ab.txt
  MGS_RASTER_BWD_FLAGS="-DMGS_RASTER_BWD_XCD_RUN=$r" SEGS=256,512 ROUNDS=5 python scripts/raster_bwd_split_ab.py 2>&1 | grep segment >> $O/ab.txt
  cd /tmp; export TMPDIR=/tmp
  for ctr in FETCH_SIZE WRITE_SIZE; do
    MGS_RASTER_BWD_FLAGS="-DMGS_RASTER_BWD_XCD_RUN=$r" MORTON=0 SEG=256 timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /root/repo/$O/pmc_${r}_$ctr -o pmc -- python /root/repo/scripts/run_stage.py raster_bwd_split 3 > /dev/null 2>&1
    python - <<PY >> /root/repo/$O/ab.txt
import csv, glob
v=[float(r["Counter_Value"]) for f in glob.glob("/root/repo/$O/pmc_${r}_$ctr/**/pmc_counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "raster_bwd_kernel" in r["Kernel_Name"]]
print("   $ctr KiB per launch:", round(sum(v[1:])/max(1,len(v[1:]))))
PY
  done
  cd /root/repo
done
python robosimgs_amd/csrc/build.py > /dev/null 2>&1
cat $O/ab.txt
