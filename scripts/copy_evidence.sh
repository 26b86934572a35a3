#!/bin/bash
# After `gpurun -- bash scripts/refresh_evidence.sh`: copy what profiles/ quotes from gpurun_out/refresh/ (run from anywhere).
cd "$(dirname "$0")/.." || exit 1
R=gpurun_out/refresh; P=profiles/r6
cp $R/refresh_default_line.json $P/03_bench_line.json
cp $R/refresh_default_stats.md $P/03_kernel_stats_default_3_in_flight.md
cp $R/bench_no_profiler.json $P/03b_bench_line_no_profiler.json
cp $R/refresh_inflight1_line.json $P/04_bench_line_inflight1.json
cp $R/refresh_inflight1_stats.md $P/04_kernel_stats_inflight1.md
cp $R/smoke.txt $P/05_smoke.txt
cp $R/06_pmc_counters.md $P/06_pmc_counters.md
cp $R/ring_world_of_one.json $P/07_bench_line_ring_world_of_one.json
cp $R/pytest_gpu.txt $P/08_pytest_gpu.txt
cp $R/bwd_timeline_0.txt $P/11_bwd_timeline_whole_list.txt
cp $R/bwd_timeline_256.txt $P/11_bwd_timeline_segments_256.txt
cp $R/bwd_split_ab.txt $P/12_bwd_split_ab.txt
cp $R/pmc_bwd_matrix.md $P/06b_pmc_backward_traffic_by_order_and_walk.md
cp $R/bench_driver_command.json $P/03c_bench_line_driver_command.json
cp $R/heavy_binning_kernels.md $P/07_heavy_tailed_binning_kernels.md
cp $R/heavy_training_step_kernels.txt $P/18_heavy_tailed_training_step_kernels_by_order.txt
cp $R/binning_kernels_at_4k.md $P/05_binning_kernels_at_4k.md
cp $R/soak_pixel_cause.txt $P/02_soak_outliers_root_cause.txt
cp $R/heavy_gates.txt $P/02_heavy_tailed_gates.txt
cp $R/sgpr_occupancy.txt $P/19d_sgpr_occupancy_microbenchmark.txt
cp $R/resident_waves_by_kernel.txt $P/19e_resident_waves_by_kernel.txt
cp $R/pmc_traffic.json profiles/pmc_traffic.json
python -c "
import json; from robosimgs_amd.csrc import build as B
print('pmc stamp current:', json.load(open('profiles/pmc_traffic.json'))['stamp'] == B.current_stamp())"
