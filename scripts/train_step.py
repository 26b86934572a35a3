"""The training step bench.py reports (configs[2], RGB+ED, fused L1, one HIP graph), alone: ms per step.
MORTON=1 times it on the Morton-ordered copy of the scene instead of the order it is given in."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, rasterization, l1_loss
n, W, H, deg = 1_000_000, 1920, 1080, 3
dev = "cuda"
g = synthetic_scene(n, math.log(0.012), deg, 0)
if os.environ.get("MORTON", "0") != "0":
    g = g.sorted_by_locality()
cam = camera_ring(1, W, H, thetas=[0.3])[0]
t = g.to_torch(dev, deg)
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)[None]
K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)[None]
params = {k: t[k].detach().clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
target = torch.rand(1, H, W, 4, device=dev, generator=torch.Generator(dev).manual_seed(1))
def step():
    for p in params.values(): p.grad = None
    c, a, meta = rasterization(params["means"], params["quats"], params["scales"], params["opacities"], params["colors"], vm, K, W, H, sh_degree=deg, render_mode="RGB+ED", isect_capacity=4_700_000)
    l1_loss(c, target).backward()
for _ in range(3): step()
torch.cuda.synchronize()
side = torch.cuda.Stream(dev)
with torch.cuda.stream(side):
    step(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        step()
torch.cuda.synchronize()
for _ in range(5): gr.replay()
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(30): gr.replay()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 30)
print(f"{os.environ.get('TAG', '')} training step (hip graph): {best * 1e3:.4f} ms   grad checksum {float(params['means'].grad.double().abs().sum()):.6e}")
