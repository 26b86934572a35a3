"""Work counters of the forward raster (instrumented build; SURVEY.md 8(d) pair-evaluation counts).

    MGS_RASTER_FWD_FLAGS=-DMGS_RASTER_STATS python robosimgs_amd/csrc/build.py && python scripts/raster_stats.py
"""
import ctypes
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robosimgs_amd import _lib, camera_ring, rasterization, synthetic_scene  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
    mu = float(sys.argv[4]) if len(sys.argv) > 4 else math.log(0.012)
    lib = _lib.lib()
    read = lib.mgs_debug_read_raster_stats
    read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    dev = torch.device("cuda", 0)
    t = synthetic_scene(n, mu, 3, seed=0).to_torch(dev, 3)
    cam = camera_ring(1, W, H, thetas=[0.3])[0]
    vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)[None]
    K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)[None]
    out = (ctypes.c_ulonglong * 8)()
    torch.cuda.synchronize()
    read(out)
    _, _, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, K,
                               W, H, sh_degree=3, render_mode="RGB+ED", raster_schedule="throughput")
    torch.cuda.synchronize()
    read(out)
    v = [int(x) for x in out]
    n_isect = int(meta["n_isects"][0])
    names = ["list entries fetched", "entries queued after cull", "quadrant evaluations",
             "lanes with valid alpha (open pixels)", "lanes accumulated", "batches processed", "batches in lists"]
    print(f"n_isect {n_isect}")
    for k, x in zip(names, v):
        print(f"{k:40s} {x:>14,d}")
    print(f"fetched / n_isect                 {v[0] / n_isect:.3f}")
    print(f"queued / fetched                  {v[1] / max(v[0], 1):.3f}")
    print(f"quadrants per queued entry        {v[2] / max(v[1], 1):.3f}")
    print(f"lane utilisation (valid / (quadrant evals*64))  {v[3] / max(v[2] * 64, 1):.3f}")


if __name__ == "__main__":
    main()
