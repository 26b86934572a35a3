cd /root/repo
for fl in "" "-DMGS_TSORT_UNIT=1024" "-DMGS_TSORT_UNIT=2048" "-DMGS_TSORT_ADAPT=96" "-DMGS_TSORT_LONG_BUCKETS=1024" "-DMGS_TSORT_UNIT=2048 -DMGS_TSORT_LONG_BUCKETS=1024"; do
  MGS_TILE_SORT_FLAGS="$fl" python robosimgs_amd/csrc/build.py --force > /dev/null 2>&1
  echo "## flags: $fl"
  SCENE=heavy MGS_TILE_SORT_FLAGS="$fl" bash scripts/prof_stage.sh binning 10 2>&1 | grep "sort\|collect" | cut -c1-100
done
