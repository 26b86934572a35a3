#!/bin/bash
# round 3: re-tune the scheduling knobs under the new kernel balance (frames/s with three frames in flight, one box)
b() { timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['single_frame_latency_ms'])"; }
echo "base"; python robosimgs_amd/csrc/build.py --force >/dev/null 2>&1; b
echo "MGS_USE_DEBUG_LIB=1 MGS_RASTER_OPTS=2 (no issue priority)"; MGS_USE_DEBUG_LIB=1 MGS_RASTER_OPTS=2 b
for f in "-DMGS_RASTER_WG_WAVES=2" "-DMGS_RASTER_WG_WAVES=8" "-DMGS_RASTER_WG_WAVES=1"; do echo "raster $f"; MGS_RASTER_FWD_FLAGS="$f" python robosimgs_amd/csrc/build.py --force >/dev/null 2>&1; MGS_RASTER_FWD_FLAGS="$f" b; done
for f in "-DMGS_DIRECT_THREADS=256" "-DMGS_DIRECT_THREADS=1024 -DMGS_DIRECT_PER_THREAD=4"; do echo "binning $f"; MGS_BINNING_FLAGS="$f" python robosimgs_amd/csrc/build.py --force >/dev/null 2>&1; MGS_BINNING_FLAGS="$f" b; done
echo "base again"; python robosimgs_amd/csrc/build.py --force >/dev/null 2>&1; b
