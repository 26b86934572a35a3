"""How long the reduce kernel's waves walk: slots per Gaussian (tight rectangles) and, per wave of 64 consecutive
Gaussians, the largest count -- the wave makes that many / 4 trips."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops
n, mu, W, H, deg, dev = 1_000_000, 0.012, 1920, 1080, 3, "cuda"
g = synthetic_scene(n, math.log(mu), deg, 0)
if os.environ.get("MORTON") == "1": g = g.sorted_by_locality()
t = g.to_torch(dev, deg)
cam = camera_ring(1, W, H, thetas=[0.3])[0]
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev); K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)
tw, th = -(-W // 16), -(-H // 16)
radii, m2d, dep, con, _, feats, splats, seed = ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False, True, want_splats=True, bin_seed="tight")
tl = ops.isect_tiles_raw(m2d, radii, dep, tw, th, 4_700_000, want_tiles_per_gauss=True, seed=seed, want_pair_info=True)
c = tl.tiles_per_gauss.cpu().numpy().astype(np.int64)
print("slots", c.sum(), "Gaussians with slots", (c > 0).sum(), "mean", c[c > 0].mean(), "max", c.max())
for thr in (8, 16, 32, 64, 128):
    print(f"  count > {thr}: {(c > thr).sum()} Gaussians, {c[c > thr].sum()} slots")
w = c[: n // 64 * 64].reshape(-1, 64)
trips = np.ceil((w.max(1) + 3) / 4)
print("waves", len(w), "trips per wave: mean", trips.mean(), "max", trips.max(), " sum", trips.sum(), " ideal (sum of slots / 4 / 64)", c.sum() / 256)
for thr in (16, 32, 64):
    wc = np.minimum(w, thr)
    print(f"  if counts above {thr} were handled elsewhere: mean trips {np.ceil((wc.max(1) + 3) / 4).mean():.2f}")
