"""Dev tool: frames/s of FrameRenderer under the two radius rules (and tile bounds), renderers taking turns in one
process so that allocation order cannot favour one.  python scripts/dbg/radius_rule_ab.py [rounds]"""
import math
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from robosimgs_amd import FrameRenderer, camera_ring, rasterization, synthetic_scene  # noqa: E402

W, H, deg, MODE = 1920, 1080, 3, "RGB+ED"
dev = torch.device("cuda", 0)
scene = synthetic_scene(1_000_000, math.log(0.012), deg, seed=0)
t = scene.to_torch(dev, deg)
cam = camera_ring(1, W, H, thetas=[0.3])[0]
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)[None]
K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)[None]
variants = [("classic rule, tight", {}), ("opacity-aware rule, tight", {"radius_rule": "opacity_aware"}),
            ("opacity-aware rule, classic bounds", {"radius_rule": "opacity_aware", "tile_bounds": "classic"}),
            ("classic rule, classic bounds", {"tile_bounds": "classic"})]
frs = []
for name, kw in variants:
    _, _, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, K, W, H, sh_degree=deg,
                               render_mode=MODE, **kw)
    n_isect = int(meta["n_isects"][0])
    n_vis = int((meta["radii"].reshape(1, len(scene), -1)[..., 0] > 0).sum())
    fr = FrameRenderer(t, W, H, render_mode=MODE, frames_in_flight=3, isect_capacity=int(5_100_000 * 1.25), **kw)
    frs.append((name, fr, n_isect, n_vis))
cd = FrameRenderer.pack_camera(vm[0].contiguous(), K[0].contiguous())


def run(fr, frames=300):
    tickets = []
    for _ in range(30):
        tk = fr.submit(cd); fr.fetch(tk, check=False); fr.release(tk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        if len(tickets) == 3:
            tk = tickets.pop(0); fr.fetch(tk, check=False); fr.release(tk)
        tickets.append(fr.submit(cd))
    while tickets:
        tk = tickets.pop(0); fr.fetch(tk, check=False); fr.release(tk)
    torch.cuda.synchronize()
    return frames / (time.perf_counter() - t0)


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
res = {name: [] for name, *_ in frs}
for _ in range(rounds):
    for name, fr, *_ in frs:
        res[name].append(run(fr))
for name, fr, n_isect, n_vis in frs:
    print(f"{name:40s} n_vis {n_vis:8d} n_isect {n_isect:8d}  frames/s " + " ".join(f"{x:7.0f}" for x in res[name]))
