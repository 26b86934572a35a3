"""(profiles/r5/00_experiments.md 20, 22)  ms per training step (forward RGB+ED + L1 + backward, one HIP graph) on four scenes: configs[2] as given / Morton,
heavy-tailed as given / Morton."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (run from scripts/)
sys.path.insert(0, ROOT)
import numpy as np, torch
from robosimgs_amd import synthetic_scene, synthetic_scene_heavy_tailed, camera_ring, rasterization, l1_loss
dev = torch.device("cuda")
W, H, deg = 1920, 1080, 3
cam = camera_ring(1, W, H, thetas=[0.3])[0]
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)[None]
K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)[None]
target = torch.rand(1, H, W, 4, device=dev, generator=torch.Generator(dev).manual_seed(1))
def step_ms(g):
    t = g.to_torch(dev, deg)
    names = ("means", "quats", "scales", "opacities", "colors")
    params = {k: t[k].detach().clone().requires_grad_(True) for k in names}
    with torch.no_grad():
        _, _, meta = rasterization(*[params[k] for k in names], vm, K, W, H, sh_degree=deg, render_mode="RGB+ED")
    cap = int(int(meta["n_isects"][0]) * 1.15) + 4096
    def train_step():
        for p in params.values(): p.grad = None
        colors, alphas, _ = rasterization(*[params[k] for k in names], vm, K, W, H, sh_degree=deg, render_mode="RGB+ED", isect_capacity=cap)
        l1_loss(colors, target).backward()
    for _ in range(3): train_step()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        train_step(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side): train_step()
    torch.cuda.synchronize()
    for _ in range(5): gr.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(30): gr.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 30 * 1e3)
    return float(np.median(ts))
iid = synthetic_scene(1_000_000, math.log(0.012), deg, 0)
hv = synthetic_scene_heavy_tailed(1_000_000, sh_degree=deg, seed=0)
for name, g in (("configs[2] as given", iid), ("configs[2] Morton", iid.sorted_by_locality()), ("heavy-tailed as given", hv), ("heavy-tailed Morton", hv.sorted_by_locality())):
    print(f"{name}: {step_ms(g):.4f} ms per step", flush=True)
