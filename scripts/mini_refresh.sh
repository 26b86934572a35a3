#!/bin/bash
# The part of refresh_evidence.sh that bench.py / oracle / test changes invalidate (no kernel changed: the PMC record stands):
#   gpurun --timeout 3000 -- bash scripts/mini_refresh.sh ; bash scripts/copy_evidence.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/refresh
mkdir -p $OUT; cd $REPO
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
python bench.py > $OUT/bench_no_profiler.json 2> $OUT/bench_no_profiler.err
python bench.py --gpus 1 --steps 20 --warmup 5 2> /dev/null | grep '^{"metric"' | tail -1 > $OUT/bench_driver_command.json
bash scripts/profile_bench.sh refresh_default
bash scripts/profile_bench.sh refresh_inflight1 --inflight 1 --no-cpu-baseline --no-stress
cp gpurun_out/refresh_default_* gpurun_out/refresh_inflight1_* $OUT/ 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_heavy.py -q -s 2>&1 | grep -v "^$" | grep "^seed\|^heavy\|passed\|failed" > $OUT/heavy_gates.txt
timeout 600 python scripts/dbg/soak_pixel_cause.py 48 67 2>&1 | grep -v amdgpu.ids > $OUT/soak_pixel_cause.txt
bash scripts/heavy_training_step_kernels.sh 2>&1 | grep -v "rocprofv3\]" > $OUT/heavy_training_step_kernels.txt
cat $OUT/pytest_gpu.txt; tail -2 $OUT/smoke.txt
