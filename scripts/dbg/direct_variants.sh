# rebuild with different workgroup sizes of the direct binning kernels; time the stage alone and the 3-in-flight bench
for v in "-DMGS_DIRECT_THREADS=256 -DMGS_DIRECT_PER_THREAD=8" "-DMGS_DIRECT_THREADS=256 -DMGS_DIRECT_PER_THREAD=16" "-DMGS_DIRECT_THREADS=512 -DMGS_DIRECT_PER_THREAD=8" "-DMGS_DIRECT_THREADS=512 -DMGS_DIRECT_PER_THREAD=4" "-DMGS_DIRECT_THREADS=512 -DMGS_DIRECT_PER_THREAD=2"; do
  MGS_BINNING_FLAGS="$v" python -c "from robosimgs_amd.csrc import build; build.build(force=True)" > /dev/null 2>&1
  TAG="[$v]" MGS_BINNING_FLAGS="$v" python scripts/binning_ab.py 2>&1 | tail -1
  MGS_BINNING_FLAGS="$v" python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   frames/s', d['value'], 'single', d['config']['single_frame_latency_ms'], 'fwd_bwd', d['fwd_bwd']['ms_per_step'])"
done
