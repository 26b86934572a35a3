# The units' kernel under its build knobs on the clustered scene: rocprofv3 kernel averages of the per-tile sort's three kernels
# (rebuilds csrc/tile_sort.hip per row on the GPU box; the shipped build is restored at the end).
#   gpurun -- bash scripts/dbg/unit_sweep.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
W6="-DMGS_TSORT_UNIT_WAVES=6"
rows=("" "-DMGS_TSORT_UNIT=1536 -DMGS_TSORT_UNIT_GRID=512" "$W6 -DMGS_TSORT_UNIT_GRID=768 -DMGS_TSORT_UNIT=2048" "$W6 -DMGS_TSORT_UNIT_GRID=768 -DMGS_TSORT_UNIT=2560"
      "$W6 -DMGS_TSORT_UNIT_GRID=768 -DMGS_TSORT_UNIT=3072" "-DMGS_TSORT_UNIT_GRID=768 -DMGS_TSORT_UNIT=2048" "$W6 -DMGS_TSORT_UNIT_GRID=640 -DMGS_TSORT_UNIT=2048"
      "$W6 -DMGS_TSORT_UNIT_GRID=768 -DMGS_TSORT_UNIT=2048 -DMGS_TSORT_ADAPT=96")
for fl in "${rows[@]}"; do
  MGS_TILE_SORT_FLAGS="$fl" python robosimgs_amd/csrc/build.py --force > /dev/null 2>&1
  echo "## flags: $fl"
  SCENE=heavy MGS_TILE_SORT_FLAGS="$fl" bash scripts/prof_stage.sh binning 10 2>&1 | grep "sort\|collect" | cut -c1-100
done
python robosimgs_amd/csrc/build.py --force > /dev/null 2>&1
