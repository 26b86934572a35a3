"""Frames/s with three frames in flight under either raster schedule (FrameRenderer(raster_schedule=...)), on configs[1], on the
heavy-tailed scene and on configs[4]: which schedule a renderer with several frames in flight should run."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robosimgs_amd import FrameRenderer, camera_ring, synthetic_scene, synthetic_scene_heavy_tailed

def fps(fr, cam_dev, frames):
    tickets = []
    def push():
        if len(tickets) == fr.n_slots:
            tk = tickets.pop(0); fr.fetch(tk, check=False); fr.release(tk)
        tickets.append(fr.submit(cam_dev))
    for _ in range(6): push()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(frames): push()
    while tickets:
        tk = tickets.pop(0); fr.fetch(tk, check=False); fr.release(tk)
    torch.cuda.synchronize()
    return frames / (time.perf_counter() - t0)

cases = [("configs[1]", lambda: synthetic_scene(1_000_000, math.log(0.012), 3, 0), 1920, 1080, 300),
         ("heavy-tailed", lambda: synthetic_scene_heavy_tailed(1_000_000, sh_degree=3, seed=0), 1920, 1080, 200),
         ("configs[4]", lambda: synthetic_scene(5_000_000, math.log(0.008), 3, 0), 3840, 2160, 90)]
for name, make, W, H, frames in cases:
    t = make().to_torch("cuda", 3)
    cam = camera_ring(1, W, H, thetas=[0.3])[0]
    cam_dev = FrameRenderer.pack_camera(torch.from_numpy(cam.viewmat().astype(np.float32)).cuda(), torch.from_numpy(cam.K.astype(np.float32)).cuda())
    rs = {}
    for sched in ("throughput", "latency"):
        rs[sched] = FrameRenderer(t, W, H, render_mode="RGB+ED", frames_in_flight=3, sizing_camera=(cam.viewmat(), cam.K), capacity_margin=1.25,
                                  raster_schedule=sched)
    rates = {k: [] for k in rs}
    for rnd in range(5):
        for k, fr in rs.items():
            rates[k].append(fps(fr, cam_dev, frames))
    print(f"{name}: " + ", ".join(f"{k} {np.median(v[1:]):.0f} frames/s" for k, v in rates.items()) + "  (3 in flight, medians of 4 interleaved rounds)")
    del rs, t
    torch.cuda.empty_cache()
