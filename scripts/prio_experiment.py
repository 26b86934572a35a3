"""Experiment: per frame, projection+binning on a high-priority stream and the raster on a normal
one (two HIP graphs per slot, event between them) against the single-graph-per-slot layout."""
import math, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops

dev = "cuda"
n, W, H, deg = 1_000_000, 1920, 1080, 3
g = synthetic_scene(n, math.log(0.012), deg, 0)
cam = camera_ring(1, W, H, thetas=[0.3])[0]
t = g.to_torch(dev, deg)
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)
K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)
tw, th = -(-W // 16), -(-H // 16)
cap = 4_700_000

def front():
    radii, means2d, depths, conics, opac, feats, splats = ops.project_color_fwd_raw(
        t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H,
        0.3, 0.01, 1e10, 0.0, False, False, want_splats=True)
    tl = ops.isect_tiles_raw(means2d, radii, depths, tw, th, cap, want_tiles_per_gauss=False,
                             conics=conics, opacities=t["opacities"])
    return means2d, conics, feats, splats, tl

def back(f, out=None):
    means2d, conics, feats, splats, tl = f
    return ops.rasterize_fwd_raw(means2d, conics, feats, t["opacities"], None, W, H, tw, th,
                                 tl.tile_offsets, tl.flatten_ids, splats=splats, track_last=False, out=out)

def capture(stream, fn):
    with torch.cuda.stream(stream):
        for _ in range(2): r = fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=stream):
            r = fn()
    torch.cuda.synchronize()
    return gr, r

for nslots in (2, 3, 4):
    for prio in (False, True):
        slots = []
        for _ in range(nslots):
            s_front = torch.cuda.Stream(priority=-1 if prio else 0)
            s_back = torch.cuda.Stream(priority=0)
            g1, f = capture(s_front, front)
            g2, o = capture(s_back, lambda: back(f))
            slots.append((s_front, s_back, g1, g2, torch.cuda.Event(), torch.cuda.Event()))
        torch.cuda.synchronize()
        iters = 300
        def run(iters):
            for i in range(iters):
                sf, sb, g1, g2, e1, e2 = slots[i % nslots]
                with torch.cuda.stream(sf):
                    sf.wait_event(e2)          # previous raster of this slot finished with the lists
                    g1.replay()
                    e1.record(sf)
                with torch.cuda.stream(sb):
                    sb.wait_event(e1)
                    g2.replay()
                    e2.record(sb)
        run(30); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(iters); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        print(f"{nslots} slots, front on {'HIGH' if prio else 'normal'} priority stream: {dt*1e3:.3f} ms/frame ({1/dt:.0f} frames/s)")
        del slots
