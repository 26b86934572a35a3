"""Dev tool: does a FrameRenderer's speed depend on when in the process it was built?  Identical renderers built one
after another (with and without deleting the previous one), each timed after its build and again at the end."""
import gc
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from robosimgs_amd import FrameRenderer, camera_ring, synthetic_scene  # noqa: E402

W, H, deg, MODE = 1920, 1080, 3, "RGB+ED"
dev = torch.device("cuda", 0)
scene = synthetic_scene(1_000_000, math.log(0.012), deg, seed=0)
t = scene.to_torch(dev, deg)
cam = camera_ring(1, W, H, thetas=[0.3])[0]
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)[None]
K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)[None]
cd = FrameRenderer.pack_camera(vm[0].contiguous(), K[0].contiguous())
CAP = int(3_708_938 * 1.25) + 4096


def run(fr, frames=600):
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


def build(**kw):
    return FrameRenderer(t, W, H, render_mode=MODE, frames_in_flight=3, isect_capacity=CAP, **kw)


print("alloc conf:", os.environ.get("PYTORCH_HIP_ALLOC_CONF"), os.environ.get("PYTORCH_CUDA_ALLOC_CONF"))
a = build(); ra = [run(a), run(a)]
b = build(); rb = [run(b), run(b)]
print(f"A (first) {ra[0]:.0f} {ra[1]:.0f}   B (second, A alive) {rb[0]:.0f} {rb[1]:.0f}   A again {run(a):.0f}")
del a; gc.collect(); torch.cuda.synchronize()
c = build(); rc = [run(c), run(c)]
print(f"C (third, A deleted first) {rc[0]:.0f} {rc[1]:.0f}   B again {run(b):.0f}")
g = build(reorder=None); rg = [run(g), run(g)]
print(f"G (caller's order, fourth) {rg[0]:.0f} {rg[1]:.0f}   C again {run(c):.0f}  B again {run(b):.0f}")
del b, c, g; gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()
d = build(); print(f"D (after empty_cache) {run(d):.0f} {run(d):.0f}")
print(torch.cuda.memory_summary(abbreviated=True)[:1500])
