"""Dev tool, second part: what about deleting a FrameRenderer slows the ones built after it?
V=1: keep the deleted renderer's graphs alive; V=2: the new renderer reuses the old one's streams; V=3: nothing deleted,
four renderers alive; V=4: new streams AND old graphs kept."""
import gc
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from robosimgs_amd import FrameRenderer, camera_ring, synthetic_scene  # noqa: E402

V = int(sys.argv[1])
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


keep = []
a = build()
print(f"V={V}  A {run(a):.0f}", [hex(s["stream"].cuda_stream) for s in a._slots])
if V == 1 or V == 4:
    keep += [s["graph"] for s in a._slots]
if V == 2:
    saved = [s["stream"] for s in a._slots]
    orig = torch.cuda.Stream
    it = iter(saved * 4)
    torch.cuda.Stream = lambda *a_, **k_: next(it)
if V == 3:
    b = build(); c = build(); d = build()
    print(f"B {run(b):.0f}  C {run(c):.0f}  D {run(d):.0f}  A again {run(a):.0f}")
    sys.exit(0)
del a; gc.collect(); torch.cuda.synchronize()
c = build()
print(f"C (after del A) {run(c):.0f} {run(c):.0f}", [hex(s["stream"].cuda_stream) for s in c._slots])
del c; gc.collect(); torch.cuda.synchronize()
e = build()
print(f"E (after del C) {run(e):.0f}", [hex(s["stream"].cuda_stream) for s in e._slots])
