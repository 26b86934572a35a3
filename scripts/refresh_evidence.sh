#!/bin/bash
# One pass over everything profiles/ quotes, on the GPU box:  gpurun --timeout 2400 -- bash scripts/refresh_evidence.sh
# Results land under gpurun_out/refresh/; copy what is to be judged into profiles/ afterwards.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/refresh
mkdir -p $OUT
cd $REPO
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
bash scripts/pmc_traffic.sh > $OUT/pmc_traffic_stdout.txt 2>&1
cp gpurun_out/pmc_r4/pmc_traffic.json profiles/pmc_traffic.json     # so that the bench lines below carry `traffic`
cp gpurun_out/pmc_r4/pmc_traffic.json $OUT/pmc_traffic.json
bash scripts/profile_bench.sh refresh_default
bash scripts/profile_bench.sh refresh_inflight1 --inflight 1 --no-cpu-baseline
MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 python bench.py --force-gather --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | grep '^{"metric"' | tail -1 > $OUT/ring_world_of_one.json
cp gpurun_out/refresh_default_* gpurun_out/refresh_inflight1_* $OUT/ 2>/dev/null
cat $OUT/pytest_gpu.txt; tail -3 $OUT/smoke.txt
