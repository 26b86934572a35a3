# The per-tile sort's main kernel under build knobs: rocprofv3 kernel averages on three scenes (rebuilds csrc/tile_sort.hip per row
# on the GPU box; the shipped build is restored at the end).   gpurun -- bash scripts/dbg/main_sweep.sh "<flags>" ["<flags>" ...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
for fl in "" "$@"; do
  MGS_TILE_SORT_FLAGS="$fl" python robosimgs_amd/csrc/build.py --force > /dev/null 2>&1
  echo "## flags: $fl"
  for sc in heavy default 4k; do
    if [ $sc = 4k ]; then export N=5000000 MU=0.008 W=3840 H=2160 CAP=30100000; else unset N MU W H CAP; fi
    [ $sc = heavy ] && export CAP=6400000
    echo "# $sc"; SCENE=$sc MGS_TILE_SORT_FLAGS="$fl" bash scripts/prof_stage.sh binning 10 2>&1 | grep "tile_depth_sort\|units\|collect" | cut -c1-100
  done
done
python robosimgs_amd/csrc/build.py --force > /dev/null 2>&1
