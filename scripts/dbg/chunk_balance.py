"""How evenly do the binning kernels' workgroups (runs of 4096 consecutive Gaussians) share the (tile, Gaussian) pairs?"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops
n, mu, W, H, deg = 1_000_000, 0.012, 1920, 1080, 3
for morton in (1, 0):
    g = synthetic_scene(n, math.log(mu), deg, 0)
    if morton: g = g.sorted_by_locality()
    cam = camera_ring(1, W, H, thetas=[0.3])[0]
    t = g.to_torch("cuda", deg)
    vm = torch.from_numpy(cam.viewmat().astype(np.float32)).cuda(); K = torch.from_numpy(cam.K.astype(np.float32)).cuda()
    out = ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False, True, want_splats=True, bin_seed="tight", lean=True)
    cnt = out[-1][0][:, 1].cpu().numpy().astype(np.int64)
    for chunk in (4096, 1024):
        m = (len(cnt) + chunk - 1) // chunk
        s = np.add.reduceat(cnt, np.arange(0, len(cnt), chunk))
        big = (cnt > 24).sum()
        print(f"morton={morton} chunk {chunk}: {m} workgroups, pairs per workgroup mean {s.mean():.0f} max {s.max()} (x{s.max() / s.mean():.1f}) p99 {np.quantile(s, 0.99):.0f}, zero-pair workgroups {(s == 0).sum()}; Gaussians with > 24 tiles {big}, largest rectangle {cnt.max()}")
