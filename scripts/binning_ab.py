"""Per-kernel timing of the binning stage at config 2 (tight lists, seeded): HIP events around
mgs_isect_tiles (back to back) -- dev tool."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops
n, mu, W, H, deg = int(os.environ.get("N", 1_000_000)), float(os.environ.get("MU", 0.012)), int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080)), 3
dev = "cuda"
g = synthetic_scene(n, math.log(mu), deg, 0)
cam = camera_ring(1, W, H, thetas=[0.3])[0]
t = g.to_torch(dev, deg)
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)
K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)
tw, th = -(-W // 16), -(-H // 16)
radii, m2d, dep, con, _, feats, splats = ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False, True, want_splats=True)
cap = 40_000_000 if n > 2_000_000 else 4_700_000
def run():
    return ops.isect_tiles_raw(m2d, radii, dep, tw, th, cap, want_tiles_per_gauss=False, conics=con, opacities=t["opacities"])
for _ in range(3): tl = run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(3):
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
import hashlib
digest = hashlib.sha256(tl.flatten_ids[:int(tl.n_isect)].cpu().numpy().tobytes() + tl.tile_offsets.cpu().numpy().tobytes()).hexdigest()[:16]
print(f"{os.environ.get('TAG', '')} binning (unseeded, back to back): {best*1e3:.1f} us  n_isect {int(tl.n_isect)}  lists sha256 {digest}")
