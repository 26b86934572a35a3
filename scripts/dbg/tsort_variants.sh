for v in "-DMGS_TSORT_FAST=2048" "-DMGS_TSORT_FAST=1024" "-DMGS_TSORT_FAST=1536"; do
  MGS_TILE_SORT_FLAGS="$v" python -c "from robosimgs_amd.csrc import build; build.build(force=True)" > /dev/null 2>&1
  TAG="[$v]" MGS_TILE_SORT_FLAGS="$v" python scripts/binning_ab.py 2>&1 | tail -1
  cd /tmp && export TMPDIR=/tmp && MGS_TILE_SORT_FLAGS="$v" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ts_prof -o s -- python $GRAFT_REPO_ROOT/scripts/binning_ab.py > /dev/null 2>&1; grep "tile_depth_sort\|scan_blocksums" /tmp/ts_prof/*/s_kernel_stats.csv /tmp/ts_prof/s_kernel_stats.csv 2>/dev/null | cut -d, -f1-4 | cut -c1-200; rm -rf /tmp/ts_prof; cd $GRAFT_REPO_ROOT
done
