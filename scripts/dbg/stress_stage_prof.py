"""configs[4] (5 M Gaussians, 3840x2160) inference-frame stages, eager, for `rocprofv3 --kernel-trace --stats`:
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stress -o s -- python scripts/dbg/stress_stage_prof.py
The scene is in the caller's order unless MORTON=1."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops
n, mu, W, H, deg, dev = 5_000_000, 0.008, 3840, 2160, 3, "cuda"
g = synthetic_scene(n, math.log(mu), deg, 0)
if os.environ.get("MORTON") == "1":
    g = g.sorted_by_locality()
t = g.to_torch(dev, deg)
cam = camera_ring(1, W, H, thetas=[0.3])[0]
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)
K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)
tw, th = -(-W // 16), -(-H // 16)
for _ in range(int(os.environ.get("FRAMES", 12))):
    radii, m2d, dep, con, _, feats, splats, seed = ops.project_color_fwd_raw(
        t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False, True,
        want_splats=True, bin_seed="tight", lean=True)
    tl = ops.isect_tiles_raw(m2d, radii, dep, tw, th, 48_000_000, want_tiles_per_gauss=False, seed=seed, want_tile_ids=False)
    ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids, splats=splats,
                          track_last=False, expected_last=True, latency=True, group_order=tl.group_order, channels=4)
torch.cuda.synchronize()
print("n_isect", int(tl.n_isect))
