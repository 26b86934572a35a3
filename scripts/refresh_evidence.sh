#!/bin/bash
# One pass over everything profiles/ quotes, on the GPU box:  gpurun --timeout 3000 -- bash scripts/refresh_evidence.sh
# Results land under gpurun_out/refresh/; copy what is to be judged into profiles/ afterwards.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/refresh
mkdir -p $OUT
cd $REPO
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
bash scripts/pmc_traffic.sh > $OUT/pmc_traffic_stdout.txt 2>&1
bash scripts/pmc_bwd_matrix.sh > $OUT/pmc_bwd_matrix.md 2>&1
cp gpurun_out/pmc_r6/pmc_traffic.json profiles/pmc_traffic.json     # so that the bench lines below carry `traffic`
cp gpurun_out/pmc_r6/pmc_traffic.json $OUT/pmc_traffic.json
python scripts/summarize_pmc.py > $OUT/pmc_summary_stdout.txt 2>&1
cp profiles/r6/06_pmc_counters.md $OUT/06_pmc_counters.md
python bench.py > $OUT/bench_no_profiler.json 2> $OUT/bench_no_profiler.err
python bench.py --gpus 1 --steps 20 --warmup 5 2> /dev/null | grep '^{"metric"' | tail -1 > $OUT/bench_driver_command.json
# round 6: the export-shaped scene's kernels (binning, training step in both orders), configs[4]'s binning, the soak outliers'
# root cause and the heavy-tailed gates with their statistics
SCENE=heavy bash scripts/prof_stage.sh binning 10 > $OUT/heavy_binning_kernels.md 2>&1
bash scripts/heavy_training_step_kernels.sh 2>&1 | grep -v "rocprofv3\]" > $OUT/heavy_training_step_kernels.txt
N=5000000 MU=0.008 W=3840 H=2160 CAP=30100000 bash scripts/prof_stage.sh binning 8 > $OUT/binning_kernels_at_4k.md 2>&1
timeout 600 python scripts/dbg/soak_pixel_cause.py 48 67 2>&1 | grep -v amdgpu.ids > $OUT/soak_pixel_cause.txt
timeout 900 python -m pytest tests/test_gpu_heavy.py -q -s 2>&1 | grep -v "^$" | grep "^seed\|^heavy\|passed\|failed" > $OUT/heavy_gates.txt
(cd scripts/ubench && hipcc --offload-arch=gfx950 -O3 -o sgpr_occupancy.bin sgpr_occupancy.hip 2> /dev/null && timeout 120 ./sgpr_occupancy.bin) > $OUT/sgpr_occupancy.txt 2>&1
bash scripts/dbg/tsort_occupancy.sh 2>&1 | grep "^4k\|^1080" > $OUT/resident_waves_by_kernel.txt
bash scripts/profile_bench.sh refresh_default
bash scripts/profile_bench.sh refresh_inflight1 --inflight 1 --no-cpu-baseline --no-stress
MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 python bench.py --force-gather --no-cpu-baseline --no-stress --steps 3 --warmup 1 2>/dev/null | grep '^{"metric"' | tail -1 > $OUT/ring_world_of_one.json
cp gpurun_out/refresh_default_* gpurun_out/refresh_inflight1_* $OUT/ 2>/dev/null
# per-unit timelines of the backward raster (instrumented build), then the shipped build again
MGS_RASTER_BWD_FLAGS="-DMGS_RASTER_BWD_TIMING" python robosimgs_amd/csrc/build.py > /dev/null 2>&1
for s in 0 256; do MGS_RASTER_BWD_FLAGS="-DMGS_RASTER_BWD_TIMING" SEG=$s timeout 300 python scripts/dbg/bwd_timeline.py 2>&1 | grep -v amdgpu.ids > $OUT/bwd_timeline_$s.txt; done
python robosimgs_amd/csrc/build.py > /dev/null 2>&1
SEGS=128,256 timeout 300 python scripts/raster_bwd_split_ab.py 2>&1 | grep segment > $OUT/bwd_split_ab.txt
cat $OUT/pytest_gpu.txt; tail -3 $OUT/smoke.txt; cat $OUT/bwd_split_ab.txt
