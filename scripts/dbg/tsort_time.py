"""Time of the seeded binning stage alone (no raster is ever run on the lists: safe for MGS_TSORT_STOP measurement builds)."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops
n, mu, W, H, deg = 1_000_000, 0.012, 1920, 1080, 3
dev = "cuda"
g = synthetic_scene(n, math.log(mu), deg, 0)
if os.environ.get("MORTON", "1") != "0":
    g = g.sorted_by_locality()
cam = camera_ring(1, W, H, thetas=[0.3])[0]
t = g.to_torch(dev, deg)
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)
K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)
tw, th = -(-W // 16), -(-H // 16)
CAP = 4_640_000
def proj():
    return ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False, True, want_splats=True, bin_seed="tight", lean=True)
dep = proj()[2]
seeds = [proj()[-1] for _ in range(24)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(4):
    fresh = [(s[0], s[1].clone()) for s in seeds[:20]]
    torch.cuda.synchronize()
    e0.record()
    for sd in fresh:
        ops.isect_tiles_raw(None, None, dep, tw, th, CAP, want_tiles_per_gauss=False, seed=sd, want_tile_ids=False, want_group_order=os.environ.get('NO_ORDER') is None)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
print(f"{os.environ.get('TAG', '')}: seeded binning stage {best * 1e3:.1f} us", flush=True)
