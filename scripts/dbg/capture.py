import math, sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, rasterization, l1_loss
mode, variant = sys.argv[1], sys.argv[2]
dev = torch.device("cuda", 0)
W, H, deg = 640, 360, 3
g = synthetic_scene(100_000, math.log(0.02), deg, 0)
cam = camera_ring(1, W, H, thetas=[0.3])[0]
t = g.to_torch(dev, deg)
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)[None]
K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)[None]
names = ("means", "quats", "scales", "opacities", "colors")
params = {k: t[k].detach().clone().requires_grad_(True) for k in names}
ch = 3 if mode == "RGB" else 4
target = torch.rand(1, H, W, ch, device=dev)
def train_step():
    for p in params.values():
        p.grad = None
    colors, alphas, meta = rasterization(params["means"], params["quats"], params["scales"], params["opacities"],
                                         params["colors"], vm, K, W, H, sh_degree=deg, render_mode=mode, isect_capacity=2_000_000)
    if variant == "l1":
        loss = l1_loss(colors, target)
    elif variant == "torch":
        loss = (colors - target).abs().mean()
    else:
        loss = (colors[..., :3] - target[..., :3]).abs().mean()
    loss.backward()
    return loss
for _ in range(3):
    train_step()
torch.cuda.synchronize()
print("eager ok", mode, variant, flush=True)
side = torch.cuda.Stream(dev)
with torch.cuda.stream(side):
    train_step()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        train_step()
torch.cuda.synchronize()
print("capture ok", flush=True)
gr.replay(); torch.cuda.synchronize()
print("replay ok", mode, variant, flush=True)
