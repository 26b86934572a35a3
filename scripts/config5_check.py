"""BASELINE configs[4]: 5 M Gaussians, SH3, 3840x2160 forward; compared against the C++ port
on image statistics + full-image diff (the CPU port takes a few seconds on the box's cores)."""
import math, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring
from robosimgs_amd.rendering import rasterization, check_isect_status
from oracle import cpu_ref
n, mu, W, H, deg = 5_000_000, 0.008, 3840, 2160, 3
g = synthetic_scene(n, math.log(mu), deg, 0)
cam = camera_ring(1, W, H, thetas=[0.3])[0]
t = g.to_torch("cuda", deg)
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).cuda()[None]
K = torch.from_numpy(cam.K.astype(np.float32)).cuda()[None]
c, a, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, K, W, H, sh_degree=deg)
torch.cuda.synchronize()
n_isect = int(meta["n_isects"][0]); print("n_vis", int((meta["radii"] > 0).sum()), "n_isect", n_isect, "(survey: 3,797,688 / 35,799,376)")
cap = int(n_isect * 1.1)
for _ in range(3):
    c, a, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, K, W, H, sh_degree=deg, isect_capacity=cap)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    c, a, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, K, W, H, sh_degree=deg, isect_capacity=cap)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
check_isect_status(meta)
print("gpu %.3f ms/frame, mem %.2f GB" % (dt * 1e3, torch.cuda.max_memory_allocated() / 1e9))
t0 = time.perf_counter()
ref, ra, info = cpu_ref.render(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, cam.viewmat(), cam.K, W, H, deg)
print("cpu %.2f s" % (time.perf_counter() - t0), info)
d = np.abs(c[0].cpu().numpy() - ref).max(-1); da = np.abs(a[0, ..., 0].cpu().numpy() - ra)
print("max diff rgb %.3e alpha %.3e; pixels over 1e-4: %d of %d (%.5f%%)" % (d.max(), da.max(), int(((d > 1e-4) | (da > 1e-4)).sum()), d.size, 100 * ((d > 1e-4) | (da > 1e-4)).mean()))
