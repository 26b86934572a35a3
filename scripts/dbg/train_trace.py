"""One eager training step (configs[2]) for a rocprofv3 kernel trace: which kernels / copies / fills a step launches."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, rasterization, l1_loss
n, W, H, deg = 1_000_000, 1920, 1080, 3
dev = "cuda"
g = synthetic_scene(n, math.log(0.012), deg, 0)
cam = camera_ring(1, W, H, thetas=[0.3])[0]
t = g.to_torch(dev, deg)
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)[None]
K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)[None]
names = ("means", "quats", "scales", "opacities", "colors")
params = {k: t[k].detach().clone().requires_grad_(True) for k in names}
target = torch.rand(1, H, W, 4, device=dev)
def step():
    for p in params.values(): p.grad = None
    c, a, meta = rasterization(params["means"], params["quats"], params["scales"], params["opacities"], params["colors"], vm, K, W, H, sh_degree=deg, render_mode="RGB+ED", isect_capacity=4_700_000)
    l1_loss(c, target).backward()
for _ in range(3): step()
torch.cuda.synchronize()
print("MARK"); step(); torch.cuda.synchronize()
