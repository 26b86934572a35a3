"""Dev tool: frames/s against frames in flight with the slots' streams chosen by hardware queue (pipeline.py
independent_streams).  GPU_MAX_HW_QUEUES from the environment.  python scripts/dbg/inflight_sweep.py 2,3,4,5,6"""
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from robosimgs_amd import FrameRenderer, camera_ring, synthetic_scene  # noqa: E402
from robosimgs_amd import pipeline  # noqa: E402

W, H, deg, MODE = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080)), 3, "RGB+ED"
dev = torch.device("cuda", 0)
scene = synthetic_scene(int(os.environ.get("N", 1_000_000)), math.log(float(os.environ.get("MU", 0.012))), deg, seed=0)
t = scene.to_torch(dev, deg)
cam = camera_ring(1, W, H, thetas=[0.3])[0]
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)[None]
K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)[None]
cd = FrameRenderer.pack_camera(vm[0].contiguous(), K[0].contiguous())
CAP = int(int(os.environ.get("NISECT", 3_708_938)) * 1.25) + 4096


def run(fr, n_fl, frames=int(os.environ.get("FRAMES", 800))):
    tickets = []
    for _ in range(30):
        tk = fr.submit(cd); fr.fetch(tk, check=False); fr.release(tk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        if len(tickets) == n_fl:
            tk = tickets.pop(0); fr.fetch(tk, check=False); fr.release(tk)
        tickets.append(fr.submit(cd))
    while tickets:
        tk = tickets.pop(0); fr.fetch(tk, check=False); fr.release(tk)
    torch.cuda.synchronize()
    return frames / (time.perf_counter() - t0)


print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))
for n_fl in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2,3,4,5,6").split(",")]:
    for sched in ("throughput", "latency"):
        fr = FrameRenderer(t, W, H, render_mode=MODE, frames_in_flight=n_fl, isect_capacity=CAP, raster_schedule=sched)
        st = list(pipeline._SLOT_STREAMS.values())[0]
        r = [run(fr, n_fl) for _ in range(2)]
        print(f"in flight {n_fl} {sched:10s}: {r[0]:.0f} {r[1]:.0f} frames/s   (distinct queues found {len(st['reps'])}, streams tried {len(st['pool'])})")
        del fr
