#!/bin/bash
# On the GPU box:  gpurun --timeout 2400 -- bash scripts/final_tree_checks.sh
# The whole -m gpu suite on libmgs_debug.so with the alternatives kept in the library forced on, then the soak scripts.
# Results: gpurun_out/final/{knobs.txt, soak.txt}  (copied to profiles/r6/10_pytest_gpu_under_knobs.txt, 09_soak_final_tree.txt)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/final
mkdir -p $OUT
cd $REPO
{
  echo "# Round 6, final tree: the whole -m gpu suite on libmgs_debug.so with the alternatives kept in the library forced on"
  for kn in "MGS_SORT_OPTS=4" "MGS_SORT_OPTS=0x08" "MGS_RASTER_OPTS=11"; do
    echo "== MGS_USE_DEBUG_LIB=1 $kn"
    env MGS_USE_DEBUG_LIB=1 $kn timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -3
  done
} > $OUT/knobs.txt
{
  timeout 600 python scripts/soak_oracle.py 120 both 2>&1 | grep -v amdgpu.ids | tail -3
  timeout 600 python scripts/soak.py 2>&1 | grep -v amdgpu.ids | tail -3
  timeout 900 python scripts/soak_backward.py 2>&1 | grep -v amdgpu.ids | tail -2
} > $OUT/soak.txt
cat $OUT/knobs.txt $OUT/soak.txt
