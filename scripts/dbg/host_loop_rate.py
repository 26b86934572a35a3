"""Dev tool: how many frames per second can the HOST side of FrameRenderer's submit / fetch / release loop issue?
A scene so small that the GPU is never the limit; bench.py's loop shape."""
import math, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from robosimgs_amd import FrameRenderer, camera_ring, synthetic_scene
dev = torch.device("cuda", 0)
W, H = 64, 48
g = synthetic_scene(500, math.log(0.05), 3, 0)
t = g.to_torch(dev, 3)
cam = camera_ring(1, W, H)[0]
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)
K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)
cd = FrameRenderer.pack_camera(vm, K)
for n_fl in (3,):
    fr = FrameRenderer(t, W, H, render_mode="RGB+ED", frames_in_flight=n_fl, isect_capacity=50_000)
    for rep in range(3):
        tickets = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        N = 3000
        for _ in range(N):
            if len(tickets) == n_fl:
                tk = tickets.pop(0); fr.fetch(tk, check=False); fr.release(tk)
            tickets.append(fr.submit(cd))
        t_issue = time.perf_counter() - t0
        while tickets:
            tk = tickets.pop(0); fr.fetch(tk, check=False); fr.release(tk)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"in flight {n_fl}: host issued {N / t_issue:.0f} frames/s ({t_issue / N * 1e6:.1f} us per frame), completed {N / dt:.0f} frames/s")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
tickets = []
for _ in range(2000):
    if len(tickets) == 3:
        tk = tickets.pop(0); fr.fetch(tk, check=False); fr.release(tk)
    tickets.append(fr.submit(cd))
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
