"""BASELINE configs[4] (5 M Gaussians, SH 3, 3840x2160): per-stage forward timing, RGB+ED."""
import math, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops
n, mu, W, H, deg = 5_000_000, 0.008, 3840, 2160, 3
g = synthetic_scene(n, math.log(mu), deg, 0)
cam = camera_ring(1, W, H, thetas=[0.3])[0]
t = g.to_torch("cuda", deg)
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).cuda()
K = torch.from_numpy(cam.K.astype(np.float32)).cuda()
tw, th = -(-W // 16), -(-H // 16)
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
def frame(cap, latency, rec=None):
    e0 = ev()
    radii, m2d, dep, con, _, feats, splats, seed = ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False, True, want_splats=True, bin_seed="tight")
    e1 = ev()
    tl = ops.isect_tiles_raw(m2d, radii, dep, tw, th, cap, want_tiles_per_gauss=False, conics=con, opacities=t["opacities"], seed=seed)
    e2 = ev()
    out = ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids, splats=splats, track_last=False, expected_last=True, latency=latency)
    e3 = ev()
    if rec is not None: rec.append((e0, e1, e2, e3))
    return tl
tl = frame(40_000_000, True); torch.cuda.synchronize()
n_isect = int(tl.n_isect); lens = (tl.tile_offsets[1:] - tl.tile_offsets[:-1]).float()
print("n_isect (tight)", n_isect, "list length mean %.0f p99 %.0f max %.0f" % (lens.mean(), lens.quantile(0.99), lens.max()))
cap = int(n_isect * 1.1)
for lat in (True, False):
    rec = []
    for _ in range(3): frame(cap, lat)
    for _ in range(10): frame(cap, lat, rec)
    torch.cuda.synchronize()
    ts = np.array([[x[i].elapsed_time(x[i + 1]) for i in range(3)] for x in rec])
    print("%s: project+SH %.3f  binning %.3f  raster %.3f  total %.3f ms" % ("per-block raster" if lat else "per-tile raster ", *np.median(ts, 0), np.median(ts.sum(1))))
